"""Re-flow a Markdown file to a column limit (DESIGN.md had lines of 1-6 k characters: unreviewable diffs).

    python tools/wrap_markdown.py DESIGN.md [width=120]

Paragraphs and list items are wrapped (continuation lines indented under the item's text); fenced code blocks, headings and tables whose
rows fit the width are left alone.  A table with rows beyond the width cannot be wrapped in place (a Markdown row is one line), so it is
turned into a list: one item per row, headed by its first cell, with one `header: cell` sub-item per further non-empty column."""
import re
import sys
import textwrap


def split_row(line):
    cells = re.split(r"(?<!\\)\|", line.strip())
    if cells and cells[0].strip() == "":
        cells = cells[1:]
    if cells and cells[-1].strip() == "":
        cells = cells[:-1]
    return [c.strip().replace("\\|", "|") for c in cells]


def wrap_par(text, width, first="", rest=""):
    return textwrap.fill(" ".join(text.split()), width=width, initial_indent=first, subsequent_indent=rest, break_long_words=False, break_on_hyphens=False)


def table_to_list(rows, width):
    head = split_row(rows[0])
    out = []
    for r in rows[2:]:
        cells = split_row(r)
        if not cells:
            continue
        out.append(wrap_par(f"**{cells[0]}**" if not cells[0].startswith("**") else cells[0], width, "* ", "  "))
        for h, c in zip(head[1:], cells[1:]):
            if c:
                out.append(wrap_par(f"{h}: {c}" if h else c, width, "  - ", "    "))
    out.append("")
    return out


def main():
    path = sys.argv[1]
    width = int(sys.argv[2]) if len(sys.argv) > 2 else 120
    lines = open(path).read().split("\n")
    out, i, n = [], 0, len(lines)
    while i < n:
        ln = lines[i]
        if ln.lstrip().startswith("```"):
            out.append(ln); i += 1
            while i < n and not lines[i].lstrip().startswith("```"):
                out.append(lines[i]); i += 1
            if i < n:
                out.append(lines[i]); i += 1
            continue
        if ln.startswith("|") and i + 1 < n and re.match(r"^\|[\s:|-]+\|?\s*$", lines[i + 1]):
            j = i
            while j < n and lines[j].startswith("|"):
                j += 1
            rows = lines[i:j]
            out.extend(rows if max(len(r) for r in rows) <= width else table_to_list(rows, width))
            i = j
            continue
        if ln.startswith("#") or ln.strip() == "" or ln.startswith("<!--") or ln.startswith("{"):
            out.append(ln); i += 1
            continue
        m = re.match(r"^(\s*)([*+-]|\d+[.)])\s+", ln)
        indent = (m.group(0) if m else re.match(r"^\s*", ln).group(0))
        rest = " " * len(indent)
        par = [ln[len(indent):]]
        i += 1
        while i < n:
            nx = lines[i]
            if nx.strip() == "" or nx.startswith("#") or nx.startswith("|") or nx.lstrip().startswith("```") or re.match(r"^\s*([*+-]|\d+[.)])\s+", nx):
                break
            par.append(nx.strip()); i += 1
        out.append(wrap_par(" ".join(par), width, indent, rest))
    open(path, "w").write("\n".join(out))


if __name__ == "__main__":
    main()

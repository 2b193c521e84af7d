import sys
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
from gemm_bench import bench
for tag, a in (("ladder + 23", (210, 20100, 20100, 1, 1, 23)), ("ladder - 25", (190, 19900, 19900, 1, 1, 25)), ("ring tf", (4000, 4000, 4000, 1, 0, -1)),
               ("mo ft 13", (220, 24310 * 44, 220, 0, 1, 13)), ("mo tf 13", (220, 24310 * 44, 220, 1, 0, 13)), ("X 13", (210, 4000, 20100, 1, 1, 13))):
    d = bench(*a, tag=tag)
    print("RES %-12s %7.3f ms %5.1f TF" % (tag, d["ms"], d["tflops"]))

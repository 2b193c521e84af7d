// grouped_launch.h -- one launch for the same kernel of several fragments, or of several independent operations of one fragment (the chains of a
// parallel region, dev_region_*) (see dev_ops_hip.hip "grouped launches" and dev_ops.h dev_tape_run).
#pragma once
#include <hip/hip_runtime.h>
#include <cstring>
#include <map>
#include <new>
#include <type_traits>

namespace qemb {
constexpr int GROUP_MAX = 12;     // members of one grouped launch (more: several launches) -- six fragments x the two chains of a parallel region
template <class... A> struct Pack;
template <> struct Pack<> {
  void load(void**) {}
  template <class F, class... X> __device__ __forceinline__ void call(F&& f, const X&... x) const { f(x...); }
};
template <class H, class... T> struct Pack<H, T...> {
  H h; Pack<T...> t;
  void load(void** kp) { std::memcpy(&h, kp[0], sizeof(H)); t.load(kp + 1); }      // (bytes as recorded, padding included)
  template <class F, class... X> __device__ __forceinline__ void call(F&& f, const X&... x) const { t.call(f, x..., h); }
};
// Everything a grouped launch needs travels in the kernel-argument segment (scalar loads, no table in device memory to chase): the first
// block of every member in the concatenated grid, the members' own grids, and the members' argument packs.
// xcd != 0: member f owns the blocks b with b % 8 == f (b / 8 is its block number): workgroups are dealt round-robin over the 8 XCDs, so one member's
// workgroups share ONE XCD and its L2 -- the operands of a small fragment are fetched into one L2 instead of eight (placement is a matter of speed only:
// nothing depends on where a block really runs).
template <class P> struct GroupArgs {
  int n, xcd;
  unsigned first[GROUP_MAX], gx[GROUP_MAX], gy[GROUP_MAX], gz[GROUP_MAX];
  P tab[GROUP_MAX];
};
template <auto Body, int MAXT, class... A>
__global__ void __launch_bounds__(MAXT) grouped_kernel(const GroupArgs<Pack<A...>> a) {
  const unsigned b = blockIdx.x;
  int f = 0;
  unsigned lb;
  if (a.xcd) {
    f = (int)(b & 7u); lb = b >> 3;
    if (f >= a.n || lb >= a.gx[f] * a.gy[f] * a.gz[f]) return;
  } else {
#pragma unroll
    for (int k = 1; k < GROUP_MAX; ++k) f += (k < a.n && b >= a.first[k]) ? 1 : 0;
    lb = b - a.first[f];
  }
  const unsigned gx = a.gx[f], gy = a.gy[f];
  const uint3 bid = make_uint3(lb % gx, (lb / gx) % gy, lb / (gx * gy));
  const uint3 gdim = make_uint3(gx, gy, a.gz[f]);
  a.tab[f].call([&](const A&... x) { Body(bid, gdim, x...); });
}
struct GroupMember { void** kernel_params; unsigned gx, gy, gz; };
struct GroupInfo {
  size_t args_bytes;
  // fill the argument block of a grouped launch of n <= GROUP_MAX members; returns the number of blocks of the concatenated grid
  unsigned (*build)(void* dst, const GroupMember* members, int n);
  void (*launch)(const void* args, unsigned blocks, dim3 block, size_t lds, hipStream_t s);
};
std::map<const void*, GroupInfo>& groupable();      // dev_ops_hip.hip
bool group_xcd_mode();                                // dev_ops_hip.hip (QEMB_GROUP_XCD)
template <auto Body, int MAXT, class... A>
static void register_groupable(const void* wrapper) {
  using P = Pack<A...>;
  using GA = GroupArgs<P>;
  static_assert(std::is_trivially_copyable<P>::value, "grouped kernel arguments must be plain data");
  static_assert(sizeof(GA) <= 4000, "grouped launch arguments exceed the kernel-argument segment");
  GroupInfo gi;
  gi.args_bytes = sizeof(GA);
  gi.build = [](void* dst, const GroupMember* m, int n) -> unsigned {
    GA* a = new (dst) GA();
    a->n = n;
    a->xcd = (group_xcd_mode() && n >= 3 && n <= 8) ? 1 : 0;      // (one or two members: an XCD each would leave most of the chip idle; more than eight: no XCD each)
    unsigned first = 0, most = 0;
    for (int k = 0; k < GROUP_MAX; ++k) {
      if (k < n) { a->first[k] = first; a->gx[k] = m[k].gx; a->gy[k] = m[k].gy; a->gz[k] = m[k].gz; a->tab[k].load(m[k].kernel_params); first += m[k].gx * m[k].gy * m[k].gz; most = m[k].gx * m[k].gy * m[k].gz > most ? m[k].gx * m[k].gy * m[k].gz : most; }
      else { a->first[k] = 0xffffffffu; a->gx[k] = a->gy[k] = a->gz[k] = 1; a->tab[k] = a->tab[0]; }
    }
    return a->xcd ? 8u * most : first;
  };
  gi.launch = [](const void* args, unsigned blocks, dim3 block, size_t lds, hipStream_t s) {
    hipLaunchKernelGGL((grouped_kernel<Body, MAXT, A...>), dim3(blocks), block, lds, s, *reinterpret_cast<const GA*>(args));
  };
  groupable()[wrapper] = gi;
}
}  // namespace qemb

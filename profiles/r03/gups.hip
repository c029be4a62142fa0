// Random-access ceiling of one MI355X for the encoder's access pattern: every lane owns a region (= a ZPAQ block's model
// state) and reads / rewrites 16-byte rows at random offsets in it.  Prints giga-requests per second for several
// footprints (is it the TLB?), lane counts per wavefront (is it the per-instruction divergence?) and wavefront counts.
//   hipcc --offload-arch=gfx950 -O3 profiles/r03/gups.hip -o profiles/r03/gups && profiles/r03/gups
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) v4u g_u128;
typedef __attribute__((address_space(1))) unsigned g_u32;

// mode 0: 16-byte loads; 1: 16-byte load + store back; 2: 4-byte load + store back; 3: three 16-byte loads of one 64-byte line + one store
template <int MODE, int ILP>
__global__ __launch_bounds__(64) void gups(unsigned char* base, unsigned long long region_bytes, unsigned nregions, unsigned lanes, int iters, unsigned* sink) {
  const unsigned lane = threadIdx.x & 63u;
  const unsigned wave = blockIdx.x;
  if (lane >= lanes) return;
  const unsigned region = (wave * lanes + lane) % nregions;
  unsigned char* p = base + (unsigned long long)region * region_bytes;
  unsigned x = (wave * 64u + lane) * 2654435761u + 12345u;
  const unsigned rows = (unsigned)(region_bytes / 64);
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
    v4u v[ILP], w[ILP], u[ILP];
    unsigned off[ILP];
#pragma unroll
    for (int k = 0; k < ILP; ++k) {
      x = x * 1664525u + 1013904223u;
      off[k] = ((x >> 4) % rows) * 64u + ((x >> 28) & 3u) * 16u;
      if (MODE == 2) v[k].x = *(g_u32*)(p + off[k]);
      else if (MODE == 5) v[k] = __builtin_nontemporal_load((g_u128*)(p + off[k]));
      else if (MODE == 6 || MODE == 7) {
        const unsigned span = MODE == 6 ? 64u : 128u;
        off[k] &= ~(span - 1u);
        v[k] = *(g_u128*)(p + off[k]);
        for (unsigned q = 16; q < span; q += 16) { const v4u t = *(g_u128*)(p + off[k] + q); v[k].y += t.x; }
      }
      else v[k] = *(g_u128*)(p + off[k]);
      if (MODE == 3) { w[k] = *(g_u128*)(p + (off[k] ^ 16u)); u[k] = *(g_u128*)(p + (off[k] ^ 32u)); }
    }
#pragma unroll
    for (int k = 0; k < ILP; ++k) {
      acc += v[k].x;
      if (MODE == 3) acc += w[k].y + u[k].z;
      if (MODE == 1 || MODE == 3) { v[k].x += 1u; *(g_u128*)(p + off[k]) = v[k]; }
      if (MODE == 2) *(g_u32*)(p + off[k]) = v[k].x + 1u;
      if (MODE == 4 || MODE == 5) { v[k].x += 1u; __builtin_nontemporal_store(v[k], (g_u128*)(p + off[k])); }
      if (MODE == 6 || MODE == 7) {
        const unsigned span = MODE == 6 ? 64u : 128u;
        for (unsigned q = 0; q < span; q += 16) { v[k].x += 1u; *(g_u128*)(p + off[k] + q) = v[k]; }
      }
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE, int ILP>
double run(unsigned char* base, unsigned long long region_bytes, unsigned nregions, unsigned lanes, unsigned waves, int iters, unsigned* sink) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL((gups<MODE, ILP>), dim3(waves), dim3(64), 0, 0, base, region_bytes, nregions, lanes, iters / 8, sink);   // warm
  CK(hipEventRecord(a));
  hipLaunchKernelGGL((gups<MODE, ILP>), dim3(waves), dim3(64), 0, 0, base, region_bytes, nregions, lanes, iters, sink);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  const double accesses = (double)waves * lanes * iters * ILP;      // rows touched (mode 1/2: a load and a store each; mode 3: 3 loads + 1 store)
  return accesses / (ms * 1e-3) / 1e9;
}

int main(int argc, char** argv) {
  if (argc > 1 && !strcmp(argv[1], "--calib")) {
    // two launches with known transaction counts, for calibrating the PMC counters (FETCH_SIZE / WRITE_SIZE per random row):
    // 2048 x 64 lanes x 256 iterations x 4 = 134 217 728 random 16-byte loads, then as many load + store pairs, over 32 GiB
    unsigned char* b = nullptr; unsigned* sk = nullptr;
    CK(hipMalloc((void**)&b, (32ull << 20) * 1024)); CK(hipMemset(b, 0, (32ull << 20) * 1024)); CK(hipMalloc((void**)&sk, 64));
    hipLaunchKernelGGL((gups<0, 4>), dim3(2048), dim3(64), 0, 0, b, 32ull << 20, 1024u, 64u, 256, sk);
    hipLaunchKernelGGL((gups<1, 4>), dim3(2048), dim3(64), 0, 0, b, 32ull << 20, 1024u, 64u, 256, sk);
    CK(hipDeviceSynchronize());
    printf("calib: 134217728 loads, then 134217728 load+store pairs\n");
    return 0;
  }
  if (argc > 1 && !strcmp(argv[1], "--inflight")) {
    // rate against requests in flight: read-modify-write of random 16-byte rows over 32 GiB, 32 live lanes per wavefront,
    // 1 / 2 / 4 independent rows per lane and 512 .. 8192 wavefronts (the encoder holds ~1 800 wavefronts with 1-2 rows
    // in flight per lane)
    unsigned char* b = nullptr; unsigned* sk = nullptr;
    CK(hipMalloc((void**)&b, (32ull << 20) * 1024)); CK(hipMemset(b, 0, (32ull << 20) * 1024)); CK(hipMalloc((void**)&sk, 64));
    printf("%-10s %12s %12s %12s   (G read-modify-writes / s; rows in flight = wavefronts x 32 x per-lane)\n", "wavefronts", "1 per lane", "2 per lane", "4 per lane");
    for (unsigned waves = 512; waves <= 8192; waves *= 2) {
      const int iters = (int)(4096u * 2048u / waves);
      printf("%-10u %12.2f %12.2f %12.2f\n", waves, run<1, 1>(b, 32ull << 20, 1024u, 32u, waves, iters, sk),
             run<1, 2>(b, 32ull << 20, 1024u, 32u, waves, iters / 2, sk), run<1, 4>(b, 32ull << 20, 1024u, 32u, waves, iters / 4, sk));
      fflush(stdout);
    }
    return 0;
  }
  size_t free_b = 0, total_b = 0;
  CK(hipMemGetInfo(&free_b, &total_b));
  const unsigned nregions = 1024;
  const unsigned long long maxreg = 96ull << 20;
  unsigned char* base = nullptr;
  CK(hipMalloc((void**)&base, maxreg * nregions));
  CK(hipMemset(base, 0, maxreg * nregions));
  unsigned* sink = nullptr;
  CK(hipMalloc((void**)&sink, 64));
  printf("free %.1f GiB of %.1f GiB; arena %.1f GiB at %p\n", free_b / 1073741824.0, total_b / 1073741824.0, maxreg * nregions / 1073741824.0, (void*)base);
  const unsigned long long regs[4] = {1ull << 20, 8ull << 20, 32ull << 20, 96ull << 20};
  printf("%-34s %10s %10s %10s %10s   (G rows/s; footprint = 1024 regions)\n", "pattern", "1 GiB", "8 GiB", "32 GiB", "96 GiB");
  struct Cfg { const char* name; int mode; unsigned lanes, waves; int iters; };
  const Cfg cfgs[] = {
    {"16B load, 64 lanes, 2048 waves", 0, 64, 2048, 4096}, {"16B load, 32 lanes, 2048 waves", 0, 32, 2048, 4096},
    {"16B load, 32 lanes, 4096 waves", 0, 32, 4096, 2048}, {"16B load, 32 lanes, 8192 waves", 0, 32, 8192, 1024},
    {"16B load, 8 lanes, 8192 waves", 0, 8, 8192, 2048},
    {"16B load+store, 32 lanes, 2048 w", 1, 32, 2048, 4096}, {"16B load+store, 32 lanes, 4096 w", 1, 32, 4096, 2048},
    {"16B load+store, 64 lanes, 4096 w", 1, 64, 4096, 2048},
    {"4B load+store, 32 lanes, 4096 w", 2, 32, 4096, 2048},
    {"3x16B load 1 line + store, 32 l, 4096 w", 3, 32, 4096, 2048},
    {"16B load + nt store, 32 l, 4096 w", 4, 32, 4096, 2048},
    {"16B nt load + nt store, 32 l, 4096 w", 5, 32, 4096, 2048},
    {"64B full RMW, 32 l, 4096 w", 6, 32, 4096, 1024},
    {"128B full RMW, 32 l, 4096 w", 7, 32, 4096, 1024},
  };
  for (const Cfg& c : cfgs) {
    printf("%-34s", c.name);
    for (int r = 0; r < 4; ++r) {
      double g = 0;
      if (c.mode == 0) g = run<0, 4>(base, regs[r], nregions, c.lanes, c.waves, c.iters, sink);
      else if (c.mode == 1) g = run<1, 4>(base, regs[r], nregions, c.lanes, c.waves, c.iters, sink);
      else if (c.mode == 2) g = run<2, 4>(base, regs[r], nregions, c.lanes, c.waves, c.iters, sink);
      else if (c.mode == 3) g = run<3, 4>(base, regs[r], nregions, c.lanes, c.waves, c.iters, sink);
      else if (c.mode == 4) g = run<4, 4>(base, regs[r], nregions, c.lanes, c.waves, c.iters, sink);
      else if (c.mode == 5) g = run<5, 4>(base, regs[r], nregions, c.lanes, c.waves, c.iters, sink);
      else if (c.mode == 6) g = run<6, 4>(base, regs[r], nregions, c.lanes, c.waves, c.iters, sink);
      else g = run<7, 4>(base, regs[r], nregions, c.lanes, c.waves, c.iters, sink);
      printf(" %10.2f", g);
      fflush(stdout);
    }
    printf("\n");
  }
  // the same with an uncached (fine-grained) arena: does the request size change?
  CK(hipFree(base));
  unsigned char* ub = nullptr;
  if (hipExtMallocWithFlags((void**)&ub, maxreg * nregions, hipDeviceMallocUncached) == hipSuccess) {
    CK(hipMemset(ub, 0, maxreg * nregions));
    printf("uncached arena at %p\n", (void*)ub);
    printf("%-34s %10.2f\n", "UC 16B load, 32 l, 4096 w, 96 GiB", run<0, 4>(ub, regs[3], nregions, 32, 4096, 2048, sink));
    printf("%-34s %10.2f\n", "UC 16B load+store, 96 GiB", run<1, 4>(ub, regs[3], nregions, 32, 4096, 2048, sink));
    printf("%-34s %10.2f\n", "UC 3x16B load + store, 96 GiB", run<3, 4>(ub, regs[3], nregions, 32, 4096, 2048, sink));
    printf("%-34s %10.2f\n", "UC 128B full RMW, 96 GiB", run<7, 4>(ub, regs[3], nregions, 32, 4096, 1024, sink));
  } else printf("uncached allocation failed\n");
  return 0;
}

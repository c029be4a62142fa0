// Round 6: what is a request worth that does NOT go to HBM?  The encoder's MIX 8 table (32 KiB per block, 32 MiB for 1024
// blocks) is small enough for the L2s / the Infinity Cache; its 8 row touches per input byte are 14 % of the requests.
// If the machine's random-access bound were the HBM's, taking them away (LDS) would buy little; if it is the request
// rate of the L2 / fabric, it buys their share.  Each lane does, per iteration, NB random 16-byte read-modify-writes in
// its own 96 MiB region (the hash tables) plus NS row read-modify-writes of ROWB bytes in its own SMALL region.
//   hipcc --offload-arch=gfx950 -O3 profiles/r06/gups2.hip -o profiles/r06/gups2 && profiles/r06/gups2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) v4u g_u128;

template <int NB, int NS, int ROWB>
__global__ __launch_bounds__(64) void mixk(unsigned char* big, unsigned long long region, unsigned char* small, unsigned small_bytes,
                                           unsigned lanes, int iters, unsigned* sink) {
  const unsigned lane = threadIdx.x & 63u, wave = blockIdx.x;
  if (lane >= lanes) return;
  const unsigned r = (wave * lanes + lane) & 1023u;
  unsigned char* p = big + (unsigned long long)r * region;
  unsigned char* s = small + (unsigned long long)(wave * lanes + lane) % 1024u * small_bytes;
  unsigned x = (wave * 64u + lane) * 2654435761u + 12345u, acc = 0;
  const unsigned rows = (unsigned)(region / 64), srows = small_bytes / ROWB;
  for (int it = 0; it < iters; ++it) {
    v4u v[NB > 0 ? NB : 1];
    unsigned off[NB > 0 ? NB : 1];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      x = x * 1664525u + 1013904223u;
      off[k] = ((x >> 4) % rows) * 64u + ((x >> 28) & 3u) * 16u;
      v[k] = *(g_u128*)(p + off[k]);
    }
    v4u t[NS > 0 ? NS : 1][ROWB / 16];
    unsigned so[NS > 0 ? NS : 1];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      x = x * 1664525u + 1013904223u;
      so[k] = ((x >> 8) % srows) * ROWB;
#pragma unroll
      for (int q = 0; q < ROWB / 16; ++q) t[k][q] = *(g_u128*)(s + so[k] + 16 * q);
    }
#pragma unroll
    for (int k = 0; k < NB; ++k) { acc += v[k].x; v[k].x += 1u; *(g_u128*)(p + off[k]) = v[k]; }
#pragma unroll
    for (int k = 0; k < NS; ++k)
#pragma unroll
      for (int q = 0; q < ROWB / 16; ++q) { acc += t[k][q].y; t[k][q].x += 1u; *(g_u128*)(s + so[k] + 16 * q) = t[k][q]; }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int NB, int NS, int ROWB>
double run(unsigned char* big, unsigned long long region, unsigned char* small, unsigned small_bytes, unsigned lanes, unsigned waves, int iters, unsigned* sink) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL((mixk<NB, NS, ROWB>), dim3(waves), dim3(64), 0, 0, big, region, small, small_bytes, lanes, iters / 8, sink);
  CK(hipEventRecord(a));
  hipLaunchKernelGGL((mixk<NB, NS, ROWB>), dim3(waves), dim3(64), 0, 0, big, region, small, small_bytes, lanes, iters, sink);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  return (double)ms * 1e6 / ((double)iters);      // ns per iteration of every lane (all lanes run side by side)
}

int main() {
  const unsigned long long region = 96ull << 20;
  unsigned char *big = nullptr, *small = nullptr; unsigned* sink = nullptr;
  CK(hipMalloc((void**)&big, region * 1024)); CK(hipMemset(big, 0, region * 1024));
  const unsigned small_cap = 8u << 20;
  CK(hipMalloc((void**)&small, (size_t)small_cap * 1024)); CK(hipMemset(small, 0, (size_t)small_cap * 1024));
  CK(hipMalloc((void**)&sink, 64));
  const unsigned lanes = 32, waves = 2048; const int iters = 1024;
  const double lanes_total = (double)lanes * waves;
  auto line = [&](const char* name, double ns, int nb, int ns_) {
    printf("%-64s %9.1f ns / iteration   %6.2f G big RMW/s  %6.2f G small RMW/s\n", name, ns, nb * lanes_total / ns, ns_ * lanes_total / ns);
    fflush(stdout);
  };
  line("48 big (16 B rows in 96 GiB)", run<48, 0, 128>(big, region, small, 32768, lanes, waves, iters, sink), 48, 0);
  line("56 big", run<56, 0, 128>(big, region, small, 32768, lanes, waves, iters, sink), 56, 0);
  line("48 big + 8 small: 128 B rows in 32 KiB per lane (32 MiB)", run<48, 8, 128>(big, region, small, 32768, lanes, waves, iters, sink), 48, 8);
  line("48 big + 8 small: 64 B rows in 16 KiB per lane (16 MiB)", run<48, 8, 64>(big, region, small, 16384, lanes, waves, iters, sink), 48, 8);
  line("48 big + 8 small: 128 B rows in 8 MiB per lane (8 GiB)", run<48, 8, 128>(big, region, small, 8u << 20, lanes, waves, iters, sink), 48, 8);
  line("48 big + 8 small: 64 B rows in 4 MiB per lane (4 GiB)", run<48, 8, 64>(big, region, small, 4u << 20, lanes, waves, iters, sink), 48, 8);
  line("40 big + 16 small: 128 B rows in 8 MiB per lane", run<40, 16, 128>(big, region, small, 8u << 20, lanes, waves, iters, sink), 40, 16);
  line("40 big + 16 small: 64 B rows in 4 MiB per lane", run<40, 16, 64>(big, region, small, 4u << 20, lanes, waves, iters, sink), 40, 16);
  line("0 big + 8 small: 128 B rows in 32 KiB per lane", run<0, 8, 128>(big, region, small, 32768, lanes, waves, iters * 4, sink), 0, 8);
  line("0 big + 8 small: 128 B rows in 256 KiB per lane (256 MiB)", run<0, 8, 128>(big, region, small, 262144, lanes, waves, iters * 4, sink), 0, 8);
  return 0;
}

// How should a wave write a [pixels][128 B] fp16 activation map?  (round 6: the conv layers' output stores cost 2.0 J of a 16.2 J front-end
// call - profiles/r06_*store_ablation* - because every store instruction writes 32-byte PIECES of 32 different lines.)
// Every pattern writes the same bytes: 16 B per lane, four instructions per 32 pixels, `stride` bytes between pixels (128 = cout 64,
// 256 = cout 128 with this workgroup owning one 128-byte half, 512 = cout 256).
//   P0  today's epilogue: lane (j = L & 31, hh = L >> 5), instruction i writes unit 2 i + hh of pixel j            -> 32 quarter lines per instruction
//   P1  whole lines, adjacent lanes: instruction i writes pixel 8 i + (L >> 3), unit L & 7                          -> 8 full lines per instruction
//   P2  whole lines, lanes interleaved (what a register-only transpose would produce): pixel 8 i + (L & 7), unit (L >> 3) bit-permuted
//   P3  half lines: pixel 16 (i & 1) + (L >> 2), unit 4 (i >> 1) + (L & 3)                                           -> 16 half lines per instruction
// Build: hipcc --offload-arch=gfx950 -O3 -o store_patterns store_patterns.hip ; run: ./store_patterns <pattern> <stride> <seconds>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int P>
__global__ __launch_bounds__(256) void k(char* __restrict__ out, size_t npix32, int stride) {
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * 256) >> 6;
  for (size_t g = wave; g < npix32; g += nwaves) {     // one group = 32 consecutive pixels
    char* base = out + g * 32 * (size_t)stride;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int pix, unit;
      if (P == 0) { pix = lane & 31; unit = 2 * i + (lane >> 5); }
      else if (P == 1) { pix = 8 * i + (lane >> 3); unit = lane & 7; }
      else if (P == 2) { pix = 8 * i + (lane & 7); const int t = lane >> 3; unit = ((t >> 2) & 1) | ((t & 1) << 1) | (((t >> 1) & 1) << 2); }
      else { pix = 16 * (i & 1) + (lane >> 2); unit = 4 * (i >> 1) + (lane & 3); }
      const u4 v = {(unsigned)g, (unsigned)lane, (unsigned)i, 0x3c003c00u};
      *reinterpret_cast<u4*>(base + (size_t)pix * stride + unit * 16) = v;
    }
  }
}

int main(int argc, char** argv) {
  const int P = argc > 1 ? atoi(argv[1]) : 0, stride = argc > 2 ? atoi(argv[2]) : 128;
  const double seconds = argc > 3 ? atof(argv[3]) : 1.0;
  const size_t npix = (size_t)128 * 188 * 688;          // conv2a's output: 128 images x 188 x 688 pixels
  const size_t npix32 = npix / 32, bytes = npix32 * 32 * (size_t)stride;
  char* buf;
  if (hipMalloc(&buf, bytes) != hipSuccess) return 1;
  hipMemset(buf, 0, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto launch = [&]() {
    if (P == 0) hipLaunchKernelGGL(k<0>, dim3(2048), dim3(256), 0, 0, buf, npix32, stride);
    else if (P == 1) hipLaunchKernelGGL(k<1>, dim3(2048), dim3(256), 0, 0, buf, npix32, stride);
    else if (P == 2) hipLaunchKernelGGL(k<2>, dim3(2048), dim3(256), 0, 0, buf, npix32, stride);
    else hipLaunchKernelGGL(k<3>, dim3(2048), dim3(256), 0, 0, buf, npix32, stride);
  };
  for (int i = 0; i < 3; ++i) launch();
  hipDeviceSynchronize();
  hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms1 = 0; hipEventElapsedTime(&ms1, e0, e1);
  const int iters = (int)(seconds * 1e3 / ms1) + 1;
  hipEventRecord(e0);
  for (int i = 0; i < iters; ++i) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  const double written = (double)npix * 128;
  printf("{\"pattern\": %d, \"stride\": %d, \"launch_us\": %.1f, \"written_GB\": %.3f, \"GB_per_s\": %.0f, \"iters\": %d}\n", P, stride, ms / iters * 1e3,
         written / 1e9, written / (ms / iters * 1e-3) / 1e9, iters);
  return 0;
}

// Calibration of rocprofv3's FETCH_SIZE on the access pattern of k_nms_tile (VERDICT r05 "do this" 5; MI355X_MICROARCH.md "HBM": the
// counter reports HALF the bytes of a wide coalesced 16-B/lane streaming read on gfx950, "other access widths are uncalibrated: calibrate
// on a known byte count in your own access pattern").  Every kernel below reads a KNOWN number of bytes exactly ONCE from a buffer that is
// far larger than the 256-MiB Infinity Cache (so nothing is absorbed on-die); FETCH_SIZE / bytes is the factor to apply.
//   k_stream   : the guide's case - 16 B per lane, lane-linear, 1 KiB per wave instruction
//   k_cells    : k_nms_tile's loader - a group of 4 lanes owns one 272-byte cell record (68 floats: 64 position logits + dustbin + pad);
//                lane q reads 4 x 16 B at float offsets 16 q + 4 i, plus the scalar at offset 64; consecutive lane groups = consecutive cells
//   k_cells_halo: the same loader walking 4 x 8-cell tiles WITH their 1-cell halo ring (6 x 10 cells per tile) over a [B][47][172] grid -
//                exactly the addresses k_nms_tile<0, 4> requests (1.875 x the cells; shows how much of the re-read reaches the fabric)
// Build + run (GPU box): hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip;  rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- ./fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void k_stream(const float4* __restrict__ p, size_t n16, float* sink) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
    const float4 v = p[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 12345.678f) *sink = acc;
}
__global__ __launch_bounds__(256) void k_cells(const float* __restrict__ p, size_t ncells, float* sink) {
  float acc = 0.f;
  const int qd = threadIdx.x & 3;
  for (size_t c = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 2; c < ncells; c += ((size_t)gridDim.x * 256) >> 2) {
    const float* lp = p + c * 68;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 v = *reinterpret_cast<const float4*>(lp + qd * 16 + i * 4);
      acc += v.x + v.y + v.z + v.w;
    }
    acc += lp[64];
  }
  if (acc == 12345.678f) *sink = acc;
}
__global__ __launch_bounds__(256) void k_cells_halo(const float* __restrict__ p, int B, int Hc, int Wc, float* sink) {
  const int tiles_x = (Wc + 7) / 8, tiles_y = (Hc + 3) / 4, ntiles = B * tiles_x * tiles_y;
  const int tid = threadIdx.x, cell = tid >> 2, qd = tid & 3, cyl = cell / 10, cxl = cell % 10;
  float acc = 0.f;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    int q = t;
    const int tx = q % tiles_x; q /= tiles_x;
    const int ty = q % tiles_y, b = q / tiles_y;
    const int cy = ty * 4 - 1 + cyl, cx = tx * 8 - 1 + cxl;
    if (tid < 240 && cy >= 0 && cy < Hc && cx >= 0 && cx < Wc) {
      const float* lp = p + ((size_t)(b * Hc + cy) * Wc + cx) * 68;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 v = *reinterpret_cast<const float4*>(lp + qd * 16 + i * 4);
        acc += v.x + v.y + v.z + v.w;
      }
      acc += lp[64];
    }
  }
  if (acc == 12345.678f) *sink = acc;
}

int main() {
  const int B = 512, Hc = 47, Wc = 172;                       // 4 x the benchmark's 128 images: 1.126 GB of cell records
  const size_t ncells = (size_t)B * Hc * Wc, bytes = ncells * 68 * 4;
  float *buf, *sink;
  if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
  hipMemset(buf, 0, bytes);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(k_stream, dim3(1280), dim3(256), 0, 0, reinterpret_cast<const float4*>(buf), bytes / 16, sink);
    hipLaunchKernelGGL(k_cells, dim3(1280), dim3(256), 0, 0, buf, ncells, sink);
    hipLaunchKernelGGL(k_cells_halo, dim3(1280), dim3(256), 0, 0, buf, B, Hc, Wc, sink);
  }
  hipDeviceSynchronize();
  // bytes each kernel requests (k_cells reads 65 of a record's 68 floats; lines are 128 B, so the 12 pad bytes travel anyway)
  long halo_cells = 0;
  for (int ty = 0; ty < (Hc + 3) / 4; ++ty)
    for (int tx = 0; tx < (Wc + 7) / 8; ++tx)
      for (int cy = ty * 4 - 1; cy < ty * 4 + 5; ++cy)
        for (int cx = tx * 8 - 1; cx < tx * 8 + 9; ++cx) halo_cells += cy >= 0 && cy < Hc && cx >= 0 && cx < Wc;
  printf("{\"buffer_bytes\": %zu, \"k_stream_bytes\": %zu, \"k_cells_bytes\": %zu, \"k_cells_halo_bytes_requested\": %zu, \"halo_factor\": %.4f}\n",
         bytes, bytes, bytes, (size_t)halo_cells * B * 272, (double)halo_cells / (Hc * Wc));
  return 0;
}

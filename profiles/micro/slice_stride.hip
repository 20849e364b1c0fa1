// Calibration, not product: the slice-resident kernel's ACCESS SHAPE without its arithmetic.  One 512-thread workgroup per
// "slice" (one per compute unit, 256 of them), each looping `iters` times over its own slice: read z, w and an operand table
// (256 KiB each), write z, w back in place, 16 bytes per lane, eight accesses in flight per wave.  Slice b of every array
// starts at b x stride.  Question (profiles/exp_xcd_asymmetry.sh): odd-numbered slices of the real kernel run their memory
// phases 10-15 % slower than even ones whichever XCD they are on -- is that the 256 KiB stride (address bit 18), and does a
// padded stride remove it?
//   hipcc -O3 --offload-arch=gfx950 slice_stride.hip -o slice_stride;  ./slice_stride <pad KiB> [slices] [iters] [pause us] [who] [throttle]
// Round 4: who = 1 / 2 lets only the odd / even workgroups stream (the others exit at once): is the odd workgroups' slowness a smaller
// share of a saturated memory system (they speed up alone) or a limit of their own (they do not)?  throttle = N: the EVEN workgroups
// sleep N x 0.64 us per iteration: does giving up their share speed the odd ones up?
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(512) void k_slices(float4* z, float4* w, const float4* y, size_t stride16, int iters, int pause_ticks, long long* t, int who, int throttle) {
    if ((who == 1 && !(blockIdx.x & 1)) || (who == 2 && (blockIdx.x & 1))) { if (threadIdx.x == 0) t[blockIdx.x] = 0; return; }
    const size_t base = (size_t)blockIdx.x * stride16;
    float4* zs = z + base;
    float4* ws = w + base;
    const float4* ys = y + base;
    const int tid = threadIdx.x;
    const long long t0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        // 256 KiB = 16384 float4 per array: 32 per thread, in 4 groups of 8 accesses
        for (int g = 0; g < 4; ++g) {
            float4 a[8], b[8], c[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = tid + 512 * (8 * g + u); a[u] = zs[i]; b[u] = ws[i]; c[u] = ys[i]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = tid + 512 * (8 * g + u);
                zs[i] = make_float4(a[u].x + c[u].x * 1e-9f, a[u].y + c[u].y * 1e-9f, a[u].z + c[u].z * 1e-9f, a[u].w + c[u].w * 1e-9f);
                ws[i] = make_float4(b[u].x - c[u].x * 1e-9f, b[u].y - c[u].y * 1e-9f, b[u].z - c[u].z * 1e-9f, b[u].w - c[u].w * 1e-9f);
            }
        }
        if (throttle > 0 && !(blockIdx.x & 1)) for (int k = 0; k < throttle; ++k) __builtin_amdgcn_s_sleep(24);
        if (pause_ticks > 0) {                                  // the compute phases of the real kernel: no memory traffic
            const long long p0 = wall_clock64();
            while (wall_clock64() - p0 < pause_ticks) __builtin_amdgcn_s_sleep(8);
        }
    }
    if (tid == 0) t[blockIdx.x] = wall_clock64() - t0;
}

int main(int argc, char** argv) {
    const size_t pad_kib = argc > 1 ? (size_t)atoi(argv[1]) : 0;
    const int slices = argc > 2 ? atoi(argv[2]) : 256;
    const int iters = argc > 3 ? atoi(argv[3]) : 50;
    const int pause_us = argc > 4 ? atoi(argv[4]) : 0;
    const int who = argc > 5 ? atoi(argv[5]) : 0, throttle = argc > 6 ? atoi(argv[6]) : 0;
    const size_t stride = (256 + pad_kib) << 10, bytes = stride * slices;
    float4 *z, *w, *y; long long* t;
    CK(hipMalloc(&z, bytes)); CK(hipMalloc(&w, bytes)); CK(hipMalloc(&y, bytes)); CK(hipMalloc(&t, slices * sizeof(long long)));
    CK(hipMemset(z, 0, bytes)); CK(hipMemset(w, 0, bytes)); CK(hipMemset(y, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<long long> h(slices);
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_slices, dim3(slices), dim3(512), 0, 0, z, w, y, stride / 16, iters, pause_us * 100, t, who, throttle);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(h.data(), t, slices * sizeof(long long), hipMemcpyDeviceToHost));
        std::vector<double> ev, od;
        for (int b = 0; b < slices; ++b) ((b & 1) ? od : ev).push_back(h[b] / 100.0 / iters);
        std::sort(ev.begin(), ev.end()); std::sort(od.begin(), od.end());
        printf("{\"who\": %d, \"throttle\": %d, \"pad_KiB\": %zu, \"slices\": %d, \"iters\": %d, \"pause_us\": %d, \"rep\": %d, \"ms\": %.3f, \"TBps\": %.3f, "
               "\"us_per_iteration_even_median\": %.2f, \"us_per_iteration_odd_median\": %.2f, \"even_max\": %.2f, \"odd_max\": %.2f}\n",
               who, throttle, pad_kib, slices, iters, pause_us, rep, ms, 5.0 * 262144.0 * (who ? slices / 2 : slices) * iters / (ms * 1e-3) / 1e12,
               ev[ev.size() / 2], od[od.size() / 2], ev.back(), od.back());
    }
    return 0;
}

// Calibration, not product: what does this GPU sustain on the slice-resident kernel's traffic MIX with a trivial
// streaming kernel?  Per "iteration": read z, w (2 x 128 MiB), read a 128 MiB operand table, write z, w in place --
// the same 3 reads : 2 writes over the same 384 MiB working set (larger than the 256 MiB MALL) as one ADMM iteration
// at batch 512.  Also pure read, pure write and copy for reference.  argv[1] = MiB per array (default 128): 64 MiB x 3 is the
// slice-resident kernel's live working set (256 slices in flight), inside the 256 MiB MALL.
//   hipcc -O3 --offload-arch=gfx950 hbm_mix.hip -o hbm_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_mix(float4* z, float4* w, const float4* y, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 a = z[i], b = w[i], c = y[i];
        z[i] = make_float4(a.x + c.x * 1e-9f, a.y + c.y * 1e-9f, a.z + c.z * 1e-9f, a.w + c.w * 1e-9f);
        w[i] = make_float4(b.x - c.x * 1e-9f, b.y - c.y * 1e-9f, b.z - c.z * 1e-9f, b.w - c.w * 1e-9f);
    }
}
// the same with four independent float4 triples in flight per thread (more bytes in flight per wave)
__global__ __launch_bounds__(256) void k_mix4(float4* z, float4* w, const float4* y, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += 4 * stride) {
        float4 a[4], b[4], c[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) if (i + u * stride < n) { a[u] = z[i + u * stride]; b[u] = w[i + u * stride]; c[u] = y[i + u * stride]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) if (i + u * stride < n) {
            z[i + u * stride] = make_float4(a[u].x + c[u].x * 1e-9f, a[u].y + c[u].y * 1e-9f, a[u].z + c[u].z * 1e-9f, a[u].w + c[u].w * 1e-9f);
            w[i + u * stride] = make_float4(b[u].x - c[u].x * 1e-9f, b[u].y - c[u].y * 1e-9f, b[u].z - c[u].z * 1e-9f, b[u].w - c[u].w * 1e-9f);
        }
    }
}
__global__ __launch_bounds__(256) void k_read(const float4* z, float* out, size_t n) {
    float s = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 a = z[i];
        s += a.x + a.y + a.z + a.w;
    }
    if (s == 123.456f) out[0] = s;
}
__global__ __launch_bounds__(256) void k_write(float4* z, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) z[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ __launch_bounds__(256) void k_copy(const float4* z, float4* w, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) w[i] = z[i];
}

int main(int argc, char** argv) {
    const size_t mib = argc > 1 ? (size_t)atoi(argv[1]) : 128;       // MiB per array (three arrays)
    const size_t bytes = mib << 20, n = bytes / 16;
    const int reps = 50;
    float4 *z, *w, *y; float* out;
    CK(hipMalloc(&z, bytes)); CK(hipMalloc(&w, bytes)); CK(hipMalloc(&y, bytes)); CK(hipMalloc(&out, 4));
    CK(hipMemset(z, 0, bytes)); CK(hipMemset(w, 0, bytes)); CK(hipMemset(y, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("{\"MiB_per_array\": %zu, ", mib);
    const int grids[] = {256 * 4, 256 * 8, 256 * 16, 256 * 32};
    for (int gi = 0; gi < 4; ++gi) {
        const int g = grids[gi];
        struct { const char* name; double gb; } rows[5] = {{"mix_3r2w", 5 * bytes / 1e9}, {"read", 3 * bytes / 1e9}, {"write", 2 * bytes / 1e9}, {"copy", 2 * bytes / 1e9}, {"mix_3r2w_4deep", 5 * bytes / 1e9}};
        for (int k = 0; k < 5; ++k) {
            for (int r = -5; r < reps; ++r) {
                if (r == 0) CK(hipEventRecord(e0));
                if (k == 0) hipLaunchKernelGGL(k_mix, dim3(g), dim3(256), 0, 0, z, w, y, n);
                if (k == 1) { hipLaunchKernelGGL(k_read, dim3(g), dim3(256), 0, 0, z, out, n); hipLaunchKernelGGL(k_read, dim3(g), dim3(256), 0, 0, w, out, n); hipLaunchKernelGGL(k_read, dim3(g), dim3(256), 0, 0, y, out, n); }
                if (k == 2) { hipLaunchKernelGGL(k_write, dim3(g), dim3(256), 0, 0, z, n); hipLaunchKernelGGL(k_write, dim3(g), dim3(256), 0, 0, w, n); }
                if (k == 3) hipLaunchKernelGGL(k_copy, dim3(g), dim3(256), 0, 0, z, w, n);
                if (k == 4) hipLaunchKernelGGL(k_mix4, dim3(g), dim3(256), 0, 0, z, w, y, n);
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%s\"%s_grid%d_TBps\": %.3f", (gi || k) ? ", " : "", rows[k].name, g, rows[k].gb * reps / ms);
        }
    }
    printf("}\n");
    return 0;
}

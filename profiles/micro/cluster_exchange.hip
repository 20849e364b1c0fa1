// Calibration, not product: what would the data movement of a "slice-resident" 512x512 kernel cost?  
//
// A 512x512 real slice is 1 MiB; a workgroup holds 256 KiB of field in its registers, so a slice needs a CLUSTER of four
// workgroups (= four compute units), and each of the two transpositions of an ADMM iteration becomes an exchange between
// them: every workgroup sends 3 x 64 KiB to its partners and receives 3 x 64 KiB.  This program runs exactly that traffic,
// without the arithmetic, on 256 resident workgroups (64 clusters):
//   per "iteration":  state phase   read z, w (2 x 256 KiB per workgroup), write them back, read 256 KiB of operand table
//                     exchange 1    write 3 pieces to the partners' inboxes (buffer A), signal, wait for all four, read 3 pieces
//                     pause         memory-free busy wait (the transforms of the real kernel)
//                     exchange 2    the same through buffer B (two buffers: seeing everybody's B-signal proves they are done with A)
//                     pause
// Hand-off protocol of MI355X_MICROARCH.md (sc1 row): every store and load of exchanged bytes carries sc1, every storing wave
// drains vmcnt, a workgroup barrier, ONE lane adds to the cluster's counter (agent scope), the consumer's lane 0 polls it with
// sc1 loads, a workgroup barrier, then the loads.  Every spin is bounded: a cluster that cannot complete raises `fail` and
// every workgroup runs to the end without waiting again.
//   hipcc -O3 --offload-arch=gfx950 cluster_exchange.hip -o cluster_exchange
//   ./cluster_exchange <mode> <iters> <pause_us> <same_xcd>     mode: 0 state only, 1 exchange only, 2 both
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t bufrsrc;
__device__ __forceinline__ bufrsrc rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
constexpr int SC1 = 16;
constexpr int PIECE = 65536;                 // bytes one workgroup sends to one partner per exchange

struct Args {
    float4 *z, *w;
    const float4* y;
    char *boxA, *boxB;                       // [cluster][receiver 4][sender 4][PIECE]
    unsigned* counters;                      // [cluster][2] on separate 128-byte lines
    unsigned* fail;
    long long* t;                            // [block][4]: state, exchange (incl. waits), pause, total  (100 MHz ticks)
    int mode, iters, pause_ticks, same_xcd;
};

__device__ __forceinline__ void exchange(const Args& a, char* box, unsigned* counter, int cluster, int me, unsigned target, u32x4 (&v)[8]) {
    const int tid = threadIdx.x;
    // send: my piece for partner p goes to inbox [cluster][p][me]
#pragma unroll
    for (int p = 1; p < 4; ++p) {
        const int to = (me + p) & 3;
        const bufrsrc r = rsrc(box + (((size_t)cluster * 4 + to) * 4 + me) * PIECE, PIECE);
#pragma unroll
        for (int u = 0; u < 8; ++u) __builtin_amdgcn_raw_buffer_store_b128(v[u], r, 16 * (tid + 512 * u), 0, SC1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(4);
            if (++spins > 4000000u || __hip_atomic_load(a.fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {   // bounded: never a hang
                __hip_atomic_store(a.fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();
    // receive: the three pieces the partners put into my inbox
#pragma unroll
    for (int p = 1; p < 4; ++p) {
        const int from = (me + p) & 3;
        const bufrsrc r = rsrc(box + (((size_t)cluster * 4 + me) * 4 + from) * PIECE, PIECE);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(r, 16 * (tid + 512 * u), 0, SC1);
            v[u].x ^= q.x; v[u].y += q.y; v[u].z ^= q.z; v[u].w += q.w;
        }
    }
}

__global__ __launch_bounds__(512) void k_cluster(Args a) {
    __shared__ char hold[96 * 1024];                         // one workgroup per compute unit, as the real kernel would be
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) hold[0] = 0;
    // same_xcd: blocks b, b+8, b+16, b+24 form a cluster (round-robin dispatch puts them on one XCD); else 4 consecutive blocks
    const int cluster = a.same_xcd ? (b / 32) * 8 + (b % 8) : b / 4;
    const int me = a.same_xcd ? (b / 8) % 4 : b % 4;
    float4* zs = a.z + (size_t)b * 16384;
    float4* ws = a.w + (size_t)b * 16384;
    const float4* ys = a.y + (size_t)b * 16384;
    unsigned* cA = a.counters + (size_t)cluster * 64;
    unsigned* cB = cA + 32;
    u32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = (u32x4){(unsigned)tid, (unsigned)u, (unsigned)b, 7u};
    long long t_state = 0, t_xchg = 0, t_pause = 0;
    const long long t0 = wall_clock64();
    for (int it = 0; it < a.iters; ++it) {
        long long s0 = wall_clock64();
        if (a.mode != 1) {
#pragma unroll 1
            for (int g = 0; g < 4; ++g) {
                float4 p[8], q[8], c[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int i = tid + 512 * (8 * g + u); p[u] = zs[i]; q[u] = ws[i]; c[u] = ys[i]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = tid + 512 * (8 * g + u);
                    zs[i] = make_float4(p[u].x + c[u].x * 1e-9f, p[u].y + c[u].y * 1e-9f, p[u].z + c[u].z * 1e-9f, p[u].w + c[u].w * 1e-9f);
                    ws[i] = make_float4(q[u].x - c[u].x * 1e-9f, q[u].y - c[u].y * 1e-9f, q[u].z - c[u].z * 1e-9f, q[u].w - c[u].w * 1e-9f);
                }
            }
        }
        long long s1 = wall_clock64();
        t_state += s1 - s0;
        for (int half = 0; half < 2; ++half) {
            if (a.mode != 0) exchange(a, half ? a.boxB : a.boxA, half ? cB : cA, cluster, me, 4u * (unsigned)(it + 1), v);
            const long long s2 = wall_clock64();
            t_xchg += s2 - s1;
            if (a.pause_ticks > 0) while (wall_clock64() - s2 < a.pause_ticks) __builtin_amdgcn_s_sleep(8);
            s1 = wall_clock64();
            t_pause += s1 - s2;
        }
    }
    if (tid == 0) {
        a.t[4 * b + 0] = t_state; a.t[4 * b + 1] = t_xchg; a.t[4 * b + 2] = t_pause; a.t[4 * b + 3] = wall_clock64() - t0;
    }
    if (v[0].x == 0x12345678u && v[3].y == 0x9abcdef0u) a.z[0].x = (float)hold[tid];        // keep the received data (and the LDS reservation) alive
}

int main(int argc, char** argv) {
    Args a{};
    a.mode = argc > 1 ? atoi(argv[1]) : 2;
    a.iters = argc > 2 ? atoi(argv[2]) : 50;
    const int pause_us = argc > 3 ? atoi(argv[3]) : 0;
    a.same_xcd = argc > 4 ? atoi(argv[4]) : 1;
    a.pause_ticks = pause_us * 100;
    const int blocks = 256, clusters = 64;
    const size_t state = (size_t)blocks * 262144, box = (size_t)clusters * 16 * PIECE;
    CK(hipMalloc(&a.z, state)); CK(hipMalloc(&a.w, state)); CK(hipMalloc((void**)&a.y, state));
    CK(hipMalloc(&a.boxA, box)); CK(hipMalloc(&a.boxB, box));
    CK(hipMalloc(&a.counters, clusters * 64 * sizeof(unsigned))); CK(hipMalloc(&a.fail, 128)); CK(hipMalloc(&a.t, blocks * 4 * sizeof(long long)));
    CK(hipMemset(a.z, 0, state)); CK(hipMemset(a.w, 0, state)); CK(hipMemset((void*)a.y, 0, state));
    CK(hipMemset(a.boxA, 0, box)); CK(hipMemset(a.boxB, 0, box));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<long long> h(blocks * 4);
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(a.counters, 0, clusters * 64 * sizeof(unsigned))); CK(hipMemset(a.fail, 0, 128));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_cluster, dim3(blocks), dim3(512), 0, 0, a);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned fail = 0;
        CK(hipMemcpy(&fail, a.fail, 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(h.data(), a.t, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
        double med[4];
        for (int k = 0; k < 4; ++k) {
            std::vector<double> v;
            for (int b = 0; b < blocks; ++b) v.push_back(h[4 * b + k] / 100.0 / a.iters);
            std::sort(v.begin(), v.end());
            med[k] = v[v.size() / 2];
        }
        const double state_bytes = a.mode != 1 ? 5.0 * 262144 : 0, xchg_bytes = a.mode != 0 ? 2.0 * 2 * 3 * PIECE : 0;   // per workgroup-iteration
        printf("{\"mode\": %d, \"iters\": %d, \"pause_us_per_half\": %d, \"same_xcd\": %d, \"rep\": %d, \"failed\": %u, \"ms\": %.3f, "
               "\"us_per_iteration\": %.2f, \"state_us\": %.2f, \"exchange_us_both\": %.2f, \"pause_us\": %.2f, "
               "\"state_MB_per_wg_iteration\": %.3f, \"exchange_MB_per_wg_iteration\": %.3f, \"aggregate_TBps\": %.3f}\n",
               a.mode, a.iters, pause_us, a.same_xcd, rep, fail, ms, med[3], med[0], med[1], med[2], state_bytes / 1e6, xchg_bytes / 1e6,
               (state_bytes + xchg_bytes) * blocks * a.iters / (ms * 1e-3) / 1e12);
    }
    return 0;
}

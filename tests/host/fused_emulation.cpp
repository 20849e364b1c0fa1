// g++ host emulation of the fused 256x256 pipeline (kernels_fused256.hip), lane by lane, using
// the SAME __host__ __device__ cores (csrc/fft16.h, csrc/fused_layout.h).  Reads a binary problem
// (two slices), runs   rows(first) -> cols -> rows(last, prox)   and writes x, z, w.
// Driven by tests/test_host_cores.py, which compares with the NumPy oracle.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../pnp_admm_cnc_mri_amd/csrc/fused_layout.h"
using namespace pnp;

static c32 TW[256];

static void coop_fft256(c32* row /*256 natural order, in place*/, bool inv) {
    // 16 lanes, lane t holds row[t + 16 j]
    static c32 lds[16 * 17];
    c32 regs[16][16];
    for (int t = 0; t < 16; ++t) {
        c32 a[16], tw[16];
        for (int j = 0; j < 16; ++j) { a[j] = row[t + 16 * j]; tw[j] = TW[t * j]; }
        if (inv) fft256_head<true>(a, tw); else fft256_head<false>(a, tw);
        for (int k = 0; k < 16; ++k) lds[k * 17 + t] = a[k];
    }
    for (int t = 0; t < 16; ++t) {
        c32 a[16];
        for (int n = 0; n < 16; ++n) a[n] = lds[t * 17 + n];
        if (inv) fft256_tail<true>(a); else fft256_tail<false>(a);
        for (int j = 0; j < 16; ++j) regs[t][j] = a[j];
    }
    for (int t = 0; t < 16; ++t) for (int j = 0; j < 16; ++j) row[t + 16 * j] = regs[t][j];
}

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    for (int m = 0; m < 256; ++m) { double a = -2.0 * M_PI * m / 256.0; TW[m] = mk((float)cos(a), (float)sin(a)); }
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 3;
    // header: int32 mode (0 = fft self-test, 1 = dc + prox), floats: c, thr, c1, c2, c3, ib, int32 cnc
    int mode, cnc; float cdc; ProxCoef pc;
    if (fread(&mode, 4, 1, f) != 1 || fread(&cnc, 4, 1, f) != 1 || fread(&cdc, 4, 1, f) != 1 || fread(&pc, sizeof(pc), 1, f) != 1) return 4;
    const int N = 65536;
    std::vector<float> z(2 * N), w(2 * N), x(2 * N);
    std::vector<c32> y(2 * N), T(N);
    std::vector<uint8_t> mask(2 * N);
    if (fread(z.data(), 4, 2 * N, f) != 2u * N || fread(w.data(), 4, 2 * N, f) != 2u * N ||
        fread(y.data(), 8, 2 * N, f) != 2u * N || fread(mask.data(), 1, 2 * N, f) != 2u * N) return 5;
    fclose(f);
    if (mode == 0) {   // transform self-test: forward FFT of rows of y[0] then inverse, dump both
        std::vector<c32> a(y.begin(), y.begin() + N), b;
        for (int r = 0; r < 256; ++r) coop_fft256(&a[r * 256], false);
        b = a;
        for (int r = 0; r < 256; ++r) coop_fft256(&b[r * 256], true);
        FILE* o = fopen(argv[2], "wb");
        fwrite(a.data(), 8, N, o); fwrite(b.data(), 8, N, o); fclose(o);
        return 0;
    }
    // ---- prepare: Hermitian tables for the pair -------------------------------------------
    std::vector<c32> Yha(YH_PAIR), Yhb(YH_PAIR);
    std::vector<uint64_t> Mh(MH_PAIR, 0);
    for (int k2 = 0; k2 <= 128; ++k2)
        for (int j = 0; j < 16; ++j)
            for (int t = 0; t < 16; ++t) {
                int k1 = t + 16 * j, ca, cb; c32 ya, yb;
                hermitian_entry(&y[0], &mask[0], k1, k2, ya, ca);
                hermitian_entry(&y[N], &mask[N], k1, k2, yb, cb);
                Yha[yh_index(0, k2, j, t)] = ya; Yhb[yh_index(0, k2, j, t)] = yb;
                Mh[mh_index(0, k2, t)] |= (uint64_t)(ca | (cb << 2)) << (4 * j);
            }
    // ---- rows (first): T[r][:] = FFT(v_a + i v_b) -------------------------------------------
    for (int r = 0; r < 256; ++r) {
        for (int n = 0; n < 256; ++n) T[r * 256 + n] = mk(z[r * 256 + n] - w[r * 256 + n], z[N + r * 256 + n] - w[N + r * 256 + n]);
        coop_fft256(&T[r * 256], false);
    }
    // ---- cols: per column pair (k2, 256-k2), lanes t = 0..15 -----------------------------------
    for (int k2 = 0; k2 <= 128; ++k2) {
        const int k2m = (256 - k2) & 255;
        std::vector<c32> Pc(256), Qc(256);
        for (int r = 0; r < 256; ++r) { Pc[r] = T[r * 256 + k2]; Qc[r] = T[r * 256 + k2m]; }
        coop_fft256(Pc.data(), false);   // P[k1] = C[k1, k2]
        coop_fft256(Qc.data(), true);    // Q[k1] = C[-k1, -k2]  (inverse-direction DFT, unscaled)
        for (int t = 0; t < 16; ++t) {
            const uint64_t code = Mh[mh_index(0, k2, t)];
            for (int j = 0; j < 16; ++j) {
                const int k1 = t + 16 * j;
                const int nib = (int)((code >> (4 * j)) & 15);
                blend_pair(Pc[k1], Qc[k1], Yha[yh_index(0, k2, j, t)], Yhb[yh_index(0, k2, j, t)], nib & 3, nib >> 2, cdc, 0.5f * cdc);
            }
        }
        coop_fft256(Pc.data(), true);    // column k2  <- inverse-direction DFT of P'
        coop_fft256(Qc.data(), false);   // column -k2 <- forward-direction DFT of Q'
        for (int r = 0; r < 256; ++r) { T[r * 256 + k2] = Pc[r]; if (k2m != k2) T[r * 256 + k2m] = Qc[r]; }
    }
    // ---- rows (last): inverse, x = |.|/65536, prox ---------------------------------------------
    const float scale = 1.0f / 65536.0f;
    for (int r = 0; r < 256; ++r) {
        coop_fft256(&T[r * 256], true);
        for (int n = 0; n < 256; ++n) {
            const c32 cv = T[r * 256 + n];
            const float xa = std::fabs(cv.x) * scale, xb = std::fabs(cv.y) * scale;
            x[r * 256 + n] = xa; x[N + r * 256 + n] = xb;
            if (cnc) { prox_cnc_pt(xa, z[r * 256 + n], w[r * 256 + n], pc); prox_cnc_pt(xb, z[N + r * 256 + n], w[N + r * 256 + n], pc); }
            else     { prox_l1_pt(xa, z[r * 256 + n], w[r * 256 + n], pc);  prox_l1_pt(xb, z[N + r * 256 + n], w[N + r * 256 + n], pc); }
        }
    }
    FILE* o = fopen(argv[2], "wb");
    fwrite(x.data(), 4, 2 * N, o); fwrite(z.data(), 4, 2 * N, o); fwrite(w.data(), 4, 2 * N, o);
    fclose(o);
    return 0;
}

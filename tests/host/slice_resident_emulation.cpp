// g++ host emulation of the slice-resident kernel (kernels_slice256.hip) with the SAME index maps
// (csrc/slice_layout.h) and cores (csrc/fft16.h), thread by thread: registers F[wave][lane][set][j],
// the two-pass LDS transposition buffer, the per-slice operand tables in thread order, the packed
// column c = 0.  One slice, one ADMM iteration from (z, w):  rows(first) -> T1 -> columns -> T2 ->
// rows(last, prox).  Input file as fused_emulation.cpp (slice 0 of it is used); output x, z, w (double).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include "../../pnp_admm_cnc_mri_amd/csrc/fused_layout.h"
#include "../../pnp_admm_cnc_mri_amd/csrc/slice_layout.h"
using namespace pnp;
typedef float R;
typedef cxT<R> C;

static C TW[256];
static C F[8][64][4][16];         // the register file: [wave][lane][set][j]
static C G[8][64][4][16];
static C LDS[SL_BUF];

// 16-lane cooperative transform of the group (wave wv, lanes 16 g .. 16 g + 15), register set `set`
static void group_fft(int wv, int g, int set, bool inv) {
    static C x[16 * 17];
    for (int t = 0; t < 16; ++t) {
        C a[16], tw[16];
        for (int j = 0; j < 16; ++j) { a[j] = F[wv][16 * g + t][set][j]; tw[j] = TW[t * j]; }
        if (inv) fft256_head<true>(a, tw); else fft256_head<false>(a, tw);
        for (int k = 0; k < 16; ++k) x[k * 17 + t] = a[k];
    }
    for (int t = 0; t < 16; ++t) {
        C a[16];
        for (int n = 0; n < 16; ++n) a[n] = x[t * 17 + n];
        if (inv) fft256_tail<true>(a); else fft256_tail<false>(a);
        for (int j = 0; j < 16; ++j) F[wv][16 * g + t][set][j] = a[j];
    }
}

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    for (int m = 0; m < 256; ++m) { const double a = -2.0 * M_PI * m / 256.0; TW[m] = mk<R>((R)cos(a), (R)sin(a)); }
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 3;
    int mode, cnc; float cdc; ProxCoef pc;
    if (fread(&mode, 4, 1, f) != 1 || fread(&cnc, 4, 1, f) != 1 || fread(&cdc, 4, 1, f) != 1 || fread(&pc, sizeof(pc), 1, f) != 1) return 4;
    const int N = 65536;
    std::vector<float> z(2 * N), w(2 * N), x(N);
    std::vector<c32> y(2 * N);
    std::vector<uint8_t> mask(2 * N);
    if (fread(z.data(), 4, 2 * N, f) != 2u * N || fread(w.data(), 4, 2 * N, f) != 2u * N ||
        fread(y.data(), 8, 2 * N, f) != 2u * N || fread(mask.data(), 1, 2 * N, f) != 2u * N) return 5;
    fclose(f);
    // ---- tables of slice 0 in column-form thread order -----------------------------------------
    std::vector<C> Yh(YH3_SLICE), Ys(256);
    std::vector<uint32_t> Mh(MH3_SLICE, 0), Ms(16, 0);
    for (int set = 0; set < SL_SETS; ++set) for (int wv = 0; wv < SL_WAVES; ++wv) for (int lane = 0; lane < 64; ++lane) {
        const int c = sl_unit(set, wv, lane), t = lane & 15;
        for (int j = 0; j < 16; ++j) {
            int code; C yh;
            hermitian_entry_t<R>(y.data(), mask.data(), t + 16 * j, c, yh, code);        // c = 0: k2 = 0
            Yh[yh3_index(0, set, j, wv, lane)] = yh;
            Mh[mh3_index(0, set, wv, lane)] |= (uint32_t)code << (2 * j);
        }
    }
    for (int t = 0; t < 16; ++t) for (int j = 0; j < 16; ++j) {
        int code; C yh;
        hermitian_entry_t<R>(y.data(), mask.data(), t + 16 * j, 128, yh, code);
        Ys[t + 16 * j] = yh;
        Ms[t] |= (uint32_t)code << (2 * j);
    }
    // ---- rows (first): F <- row transforms of the row pairs ------------------------------------
    for (int wv = 0; wv < SL_WAVES; ++wv) for (int lane = 0; lane < 64; ++lane) for (int set = 0; set < SL_SETS; ++set) {
        const int r = sl_unit(set, wv, lane), t = lane & 15;
        for (int j = 0; j < 16; ++j) {
            const int n = t + 16 * j;
            F[wv][lane][set][j] = mk<R>(z[(2 * r) * 256 + n] - w[(2 * r) * 256 + n], z[(2 * r + 1) * 256 + n] - w[(2 * r + 1) * 256 + n]);
        }
    }
    for (int wv = 0; wv < SL_WAVES; ++wv) for (int g = 0; g < 4; ++g) for (int set = 0; set < SL_SETS; ++set) group_fft(wv, g, set, false);
    // ---- T1: row form -> column form, two passes through the buffer ----------------------------
    for (int p = 0; p < 2; ++p) {
        for (int i = 0; i < SL_BUF; ++i) LDS[i] = mk<R>(NAN, NAN);
        int written = 0;
        for (int wv = 0; wv < SL_WAVES; ++wv) for (int lane = 0; lane < 64; ++lane) for (int set = 0; set < SL_SETS; ++set) {
            const int r = sl_unit(set, wv, lane), t = lane & 15;
            int mine = 0;
            for (int j = 0; j < 16; ++j) {
                const int k = t + 16 * j;
                if (sl_pass(k) != p) continue;
                if (!std::isnan(LDS[r * SL_P + sl_slot(k)].x)) return 6;          // slots must not collide
                LDS[r * SL_P + sl_slot(k)] = F[wv][lane][set][j];
                ++written; ++mine;
            }
            if (mine != 8) return 7;                                              // every thread moves 8 values per set and pass
        }
        if (written != 128 * 128) return 7;
        // the kernel's lane-level form (t1_pass): the even lane of a pair reads C[c], the odd lane C[-c]; the partner's
        // value arrives by DPP; p / q pick the components so that one add and one subtract serve both parities
        for (int wv = 0; wv < SL_WAVES; ++wv) for (int lane = 0; lane < 64; ++lane) for (int set = 2 * p; set < 2 * p + 2; ++set) {
            const int c = sl_unit(set, wv, lane), t = lane & 15, cc = c & 63, odd = t & 1;
            for (int j = 0; j < 16; ++j) {
                const int rho = t + 16 * j, r = rho >> 1;
                const C d = LDS[r * SL_P + cc], m = LDS[r * SL_P + SL_M + cc];
                if (std::isnan(d.x) || std::isnan(m.x)) return 8;
                const C own = odd ? m : d, other = odd ? d : m;                    // other = what the partner lane read
                const R pp = odd ? own.y : own.x, qq = odd ? own.x : own.y;
                const R pp_partner = odd ? other.x : other.y, qq_partner = odd ? other.y : other.x;   // the partner has the opposite parity
                C v = mk<R>((R)0.5 * (pp + qq_partner), (R)0.5 * (qq - pp_partner));
                const C ref = (rho & 1) ? unpack_b(d, m) : unpack_a(d, m);
                if (v.x != ref.x || v.y != ref.y) return 12;                       // bit-equal to unpack_a / unpack_b
                if (c == 0) {
                    v = odd ? mk<R>(qq_partner, pp) : mk<R>(pp, qq_partner);
                    const C raw = (rho & 1) ? mk<R>(d.y, m.y) : mk<R>(d.x, m.x);
                    if (v.x != raw.x || v.y != raw.y) return 13;
                }
                G[wv][lane][set][j] = v;
            }
        }
    }
    for (int wv = 0; wv < SL_WAVES; ++wv) for (int lane = 0; lane < 64; ++lane) for (int set = 0; set < SL_SETS; ++set) for (int j = 0; j < 16; ++j) F[wv][lane][set][j] = G[wv][lane][set][j];
    // ---- columns: transform, blend, inverse transform ------------------------------------------
    const R ch = 0.5f * cdc;
    for (int wv = 0; wv < SL_WAVES; ++wv) for (int g = 0; g < 4; ++g) for (int set = 0; set < SL_SETS; ++set) {
        group_fft(wv, g, set, false);
        const int c = sl_unit(set, wv, 16 * g);
        if (c == 0) {
            C Gk[256];
            for (int t = 0; t < 16; ++t) for (int j = 0; j < 16; ++j) Gk[t + 16 * j] = F[wv][16 * g + t][set][j];
            for (int t = 0; t < 16; ++t) for (int j = 0; j < 16; ++j) {
                const int k1 = t + 16 * j, lane = 16 * g + t;
                const C gk = Gk[k1], gm = Gk[(256 - k1) & 255];
                const C A = blend_one(unpack_a(gk, gm), Yh[yh3_index(0, set, j, wv, lane)], (int)((Mh[mh3_index(0, set, wv, lane)] >> (2 * j)) & 3u), cdc, ch);
                const C Bv = blend_one(unpack_b(gk, gm), Ys[k1], (int)((Ms[t] >> (2 * j)) & 3u), cdc, ch);
                F[wv][lane][set][j] = repack_p(A, Bv);
            }
        } else {
            for (int t = 0; t < 16; ++t) for (int j = 0; j < 16; ++j) {
                const int lane = 16 * g + t;
                F[wv][lane][set][j] = blend_one(F[wv][lane][set][j], Yh[yh3_index(0, set, j, wv, lane)],
                                                (int)((Mh[mh3_index(0, set, wv, lane)] >> (2 * j)) & 3u), cdc, ch);
            }
        }
        group_fft(wv, g, set, true);
    }
    // ---- T2: column form -> row form -------------------------------------------------------------
    for (int p = 0; p < 2; ++p) {
        for (int i = 0; i < SL_BUF; ++i) LDS[i] = mk<R>(NAN, NAN);
        // the kernel's lane-level form (t2_pass): A = own.x - other.y, B = own.y + other.x; the even lane writes (A, B) to the
        // direct slot, the odd lane (B, A) to the mirror slot
        for (int wv = 0; wv < SL_WAVES; ++wv) for (int lane = 0; lane < 64; ++lane) for (int set = 2 * p; set < 2 * p + 2; ++set) {
            const int c = sl_unit(set, wv, lane), t = lane & 15, cc = c & 63, odd = t & 1;
            for (int j = 0; j < 16; ++j) {
                const int r = (t + 16 * j) >> 1;
                const C own = F[wv][lane][set][j], other = F[wv][lane ^ 1][set][j];
                const R A = own.x - other.y, Bv = own.y + other.x;
                C v = odd ? mk<R>(Bv, A) : mk<R>(A, Bv);
                const C ue = odd ? other : own, uo = odd ? own : other;
                const C ref = odd ? repack_q(ue, uo) : repack_p(ue, uo);
                if (v.x != ref.x || v.y != ref.y) return 14;                       // bit-equal to repack_p / repack_q
                if (c == 0) v = odd ? mk<R>(other.y, own.y) : mk<R>(own.x, other.x);                  // = (ue.x, uo.x) / (ue.y, uo.y)
                LDS[r * SL_P + (odd ? SL_M : 0) + cc] = v;
            }
        }
        for (int wv = 0; wv < SL_WAVES; ++wv) for (int lane = 0; lane < 64; ++lane) for (int set = 0; set < SL_SETS; ++set) {
            const int r = sl_unit(set, wv, lane), t = lane & 15;
            for (int j = 0; j < 16; ++j) {
                const int k = t + 16 * j;
                if (sl_pass(k) != p) continue;
                G[wv][lane][set][j] = LDS[r * SL_P + sl_slot(k)];
                if (std::isnan(G[wv][lane][set][j].x)) return 9;
            }
        }
    }
    for (int wv = 0; wv < SL_WAVES; ++wv) for (int lane = 0; lane < 64; ++lane) for (int set = 0; set < SL_SETS; ++set) for (int j = 0; j < 16; ++j) F[wv][lane][set][j] = G[wv][lane][set][j];
    // ---- rows (last): inverse transform, x = |re|, |im| / 65536, prox --------------------------------
    for (int wv = 0; wv < SL_WAVES; ++wv) for (int g = 0; g < 4; ++g) for (int set = 0; set < SL_SETS; ++set) group_fft(wv, g, set, true);
    const R scale = 1.0f / 65536.0f;
    for (int wv = 0; wv < SL_WAVES; ++wv) for (int lane = 0; lane < 64; ++lane) for (int set = 0; set < SL_SETS; ++set) {
        const int r = sl_unit(set, wv, lane), t = lane & 15;
        for (int j = 0; j < 16; ++j) {
            const int n = t + 16 * j;
            const C o = F[wv][lane][set][j];
            const int ia = (2 * r) * 256 + n, ib = (2 * r + 1) * 256 + n;
            const R xa = std::fabs(o.x) * scale, xb = std::fabs(o.y) * scale;
            x[ia] = xa; x[ib] = xb;
            if (cnc) { prox_cnc_pt(xa, z[ia], w[ia], pc); prox_cnc_pt(xb, z[ib], w[ib], pc); }
            else     { prox_l1_pt(xa, z[ia], w[ia], pc);  prox_l1_pt(xb, z[ib], w[ib], pc); }
        }
    }
    FILE* o = fopen(argv[2], "wb");
    std::vector<double> d(3 * N);
    for (int i = 0; i < N; ++i) { d[i] = x[i]; d[N + i] = z[i]; d[2 * N + i] = w[i]; }
    fwrite(d.data(), 8, d.size(), o);
    fclose(o);
    return 0;
}

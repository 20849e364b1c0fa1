// g++ host emulation of the slice-resident kernel (kernels_slice256.hip) with the SAME index maps
// (csrc/slice_layout.h) and cores (csrc/fft16.h), thread by thread: registers F[wave][lane][set][j],
// the two-pass LDS transposition buffer, the per-slice operand tables in thread order, the packed
// column c = 0.  One slice, one ADMM iteration from (z, w):  rows(first) -> T1 -> columns -> T2 ->
// rows(last, prox).  Input file as fused_emulation.cpp (slice 0 of it is used); output x, z, w (double).
// z / w go through the kernel's HBM order of the state (sl_state_index): the emulation converts the natural-order input
// first and reads / writes every value where the kernel's lane finds it.
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
    // ---- state arrays in slice order (what api.hip's k_state_order leaves in HBM) ----------------
    std::vector<float> zs(N), ws(N);
    {
        std::vector<char> hit(N, 0);
        for (int row = 0; row < 256; ++row) for (int n = 0; n < 256; ++n) {
            const size_t at = sl_state_index(row, n);
            if (at >= (size_t)N || hit[at]) return 15;                            // the order is a permutation of the slice
            hit[at] = 1;
            zs[at] = z[row * 256 + n];
            ws[at] = w[row * 256 + n];
        }
    }
    // the lane's q-th 16-byte access of row pair r: floats 512 r + 64 q + 4 t .. + 3 = (row 2r, row 2r+1) x (j = 2q, 2q + 1)
    auto lane_access = [](int r, int t, int q, int k) { return (size_t)512 * r + 64 * q + 4 * t + k; };
    for (int r = 0; r < 128; ++r) for (int t = 0; t < 16; ++t) for (int q = 0; q < 8; ++q) for (int k = 0; k < 4; ++k)
        if (lane_access(r, t, q, k) != sl_state_index(2 * r + (k & 1), t + 16 * (2 * q + (k >> 1)))) return 16;
    // ---- rows (first): F <- row transforms of the row pairs ------------------------------------
    for (int wv = 0; wv < SL_WAVES; ++wv) for (int lane = 0; lane < 64; ++lane) for (int set = 0; set < SL_SETS; ++set) {
        const int r = sl_unit(set, wv, lane), t = lane & 15;
        for (int j = 0; j < 16; ++j) {
            const int n = t + 16 * j;
            const size_t ia = sl_state_index(2 * r, n), ib = sl_state_index(2 * r + 1, n);
            F[wv][lane][set][j] = mk<R>(zs[ia] - ws[ia], zs[ib] - ws[ib]);
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
                const C f = F[wv][lane][set][j];
                LDS[r * SL_P + sl_slot(k)] = sl_slot(k) >= SL_M ? mk<R>(f.y, f.x) : f;     // mirror half: re / im swapped
                ++written; ++mine;
            }
            if (mine != 8) return 7;                                              // every thread moves 8 values per set and pass
        }
        if (written != 128 * 128) return 7;
        // the kernel's lane-level form (t1_pass): the even lane of a pair reads C[c], the odd lane C[-c] with its halves swapped;
        // the partner's halves arrive by DPP inside one add and one subtract, the same for both parities.  The result is
        // TWICE the unpacked value (the 1/2 sits in the blend coefficients, blend_scaled)
        for (int wv = 0; wv < SL_WAVES; ++wv) for (int lane = 0; lane < 64; ++lane) for (int set = 2 * p; set < 2 * p + 2; ++set) {
            const int c = sl_unit(set, wv, lane), t = lane & 15, cc = c & 63, odd = t & 1;
            for (int j = 0; j < 16; ++j) {
                const int rho = t + 16 * j, r = rho >> 1;
                const C d = LDS[r * SL_P + cc], ms = LDS[r * SL_P + SL_M + cc];       // ms = the mirror value as stored (swapped)
                if (std::isnan(d.x) || std::isnan(ms.x)) return 8;
                const C m = mk<R>(ms.y, ms.x);
                const C own = odd ? ms : d, other = odd ? d : ms;                  // other = what the partner lane read
                C v = mk<R>(own.x + other.y, own.y - other.x);
                const C ref = (rho & 1) ? unpack_b(d, m) : unpack_a(d, m);
                if ((R)0.5 * v.x != ref.x || (R)0.5 * v.y != ref.y) return 12;    // bit-equal to 2 x unpack_a / unpack_b
                if (c == 0) {
                    v = odd ? mk<R>(other.y, own.x) : mk<R>(own.x, other.y);
                    const C raw = (rho & 1) ? mk<R>(d.y, m.y) : mk<R>(d.x, m.x);
                    if (v.x != raw.x || v.y != raw.y) return 13;
                }
                G[wv][lane][set][j] = v;
            }
        }
    }
    for (int wv = 0; wv < SL_WAVES; ++wv) for (int lane = 0; lane < 64; ++lane) for (int set = 0; set < SL_SETS; ++set) for (int j = 0; j < 16; ++j) F[wv][lane][set][j] = G[wv][lane][set][j];
    // ---- columns: transform, blend, inverse transform ------------------------------------------
    // the kernel's blend carries the inverse transforms' 1/N (blend_scaled, fft16.h): checked here, value by value, to be
    // exactly scale x blend_one, so that x = |.| below needs no multiplication
    const R ch = 0.5f * cdc, scale = 1.0f / 65536.0f, cs = cdc * scale, chs = ch * scale;
    for (int wv = 0; wv < SL_WAVES; ++wv) for (int g = 0; g < 4; ++g) for (int set = 0; set < SL_SETS; ++set) {
        group_fft(wv, g, set, false);
        const int c = sl_unit(set, wv, 16 * g);
        if (c == 0) {
            C Gk[256];
            for (int t = 0; t < 16; ++t) for (int j = 0; j < 16; ++j) Gk[t + 16 * j] = F[wv][16 * g + t][set][j];
            for (int t = 0; t < 16; ++t) for (int j = 0; j < 16; ++j) {
                const int k1 = t + 16 * j, lane = 16 * g + t;
                const C gk = Gk[k1], gm = Gk[(256 - k1) & 255];
                const C A1 = blend_one(unpack_a(gk, gm), Yh[yh3_index(0, set, j, wv, lane)], (int)((Mh[mh3_index(0, set, wv, lane)] >> (2 * j)) & 3u), cdc, ch);
                const C B1 = blend_one(unpack_b(gk, gm), Ys[k1], (int)((Ms[t] >> (2 * j)) & 3u), cdc, ch);
                const C A = blend_scaled(unpack_a(gk, gm), Yh[yh3_index(0, set, j, wv, lane)], (int)((Mh[mh3_index(0, set, wv, lane)] >> (2 * j)) & 3u), cs, chs, scale);
                const C Bv = blend_scaled(unpack_b(gk, gm), Ys[k1], (int)((Ms[t] >> (2 * j)) & 3u), cs, chs, scale);
                if (A.x != scale * A1.x || A.y != scale * A1.y || Bv.x != scale * B1.x || Bv.y != scale * B1.y) return 18;
                F[wv][lane][set][j] = repack_p(A, Bv);
            }
        } else {
            for (int t = 0; t < 16; ++t) for (int j = 0; j < 16; ++j) {
                const int lane = 16 * g + t;
                const C v2 = F[wv][lane][set][j];                                          // the doubled field
                const C got = blend_scaled(v2, Yh[yh3_index(0, set, j, wv, lane)], (int)((Mh[mh3_index(0, set, wv, lane)] >> (2 * j)) & 3u), cs, (R)0.5 * chs, (R)0.5 * scale);
                const C ref = blend_one(mk<R>((R)0.5 * v2.x, (R)0.5 * v2.y), Yh[yh3_index(0, set, j, wv, lane)],
                                        (int)((Mh[mh3_index(0, set, wv, lane)] >> (2 * j)) & 3u), cdc, ch);
                if (got.x != scale * ref.x || got.y != scale * ref.y) return 17;          // folding 1/2 and 1/N into the coefficients is exact
                F[wv][lane][set][j] = got;
            }
        }
        group_fft(wv, g, set, true);
    }
    // ---- T2: column form -> row form -------------------------------------------------------------
    for (int p = 0; p < 2; ++p) {
        for (int i = 0; i < SL_BUF; ++i) LDS[i] = mk<R>(NAN, NAN);
        // the kernel's lane-level form (t2_pass): both lanes write (own.x - other.y, own.y + other.x), the even lane to the direct
        // slot, the odd lane to the mirror slot, whose values are stored with re / im swapped
        for (int wv = 0; wv < SL_WAVES; ++wv) for (int lane = 0; lane < 64; ++lane) for (int set = 2 * p; set < 2 * p + 2; ++set) {
            const int c = sl_unit(set, wv, lane), t = lane & 15, cc = c & 63, odd = t & 1;
            for (int j = 0; j < 16; ++j) {
                const int r = (t + 16 * j) >> 1;
                const C own = F[wv][lane][set][j], other = F[wv][lane ^ 1][set][j];
                C v = mk<R>(own.x - other.y, own.y + other.x);
                const C ue = odd ? other : own, uo = odd ? own : other;
                const C ref = odd ? repack_q(ue, uo) : repack_p(ue, uo);
                if (odd ? (v.x != ref.y || v.y != ref.x) : (v.x != ref.x || v.y != ref.y)) return 14;   // bit-equal to repack_p / swapped repack_q
                if (c == 0) v = odd ? mk<R>(own.y, other.y) : mk<R>(own.x, other.x);                  // = (ue.x, uo.x) / swapped (ue.y, uo.y)
                LDS[r * SL_P + (odd ? SL_M : 0) + cc] = v;
            }
        }
        for (int wv = 0; wv < SL_WAVES; ++wv) for (int lane = 0; lane < 64; ++lane) for (int set = 0; set < SL_SETS; ++set) {
            const int r = sl_unit(set, wv, lane), t = lane & 15;
            for (int j = 0; j < 16; ++j) {
                const int k = t + 16 * j;
                if (sl_pass(k) != p) continue;
                const C got = LDS[r * SL_P + sl_slot(k)];
                if (std::isnan(got.x)) return 9;
                G[wv][lane][set][j] = sl_slot(k) >= SL_M ? mk<R>(got.y, got.x) : got;
            }
        }
    }
    for (int wv = 0; wv < SL_WAVES; ++wv) for (int lane = 0; lane < 64; ++lane) for (int set = 0; set < SL_SETS; ++set) for (int j = 0; j < 16; ++j) F[wv][lane][set][j] = G[wv][lane][set][j];
    // ---- rows (last): inverse transform, x = |re|, |im| (the 1/65536 came with the blend), prox -------
    for (int wv = 0; wv < SL_WAVES; ++wv) for (int g = 0; g < 4; ++g) for (int set = 0; set < SL_SETS; ++set) group_fft(wv, g, set, true);
    for (int wv = 0; wv < SL_WAVES; ++wv) for (int lane = 0; lane < 64; ++lane) for (int set = 0; set < SL_SETS; ++set) {
        const int r = sl_unit(set, wv, lane), t = lane & 15;
        for (int j = 0; j < 16; ++j) {
            const int n = t + 16 * j;
            const C o = F[wv][lane][set][j];
            const size_t ia = sl_state_index(2 * r, n), ib = sl_state_index(2 * r + 1, n);
            const R xa = std::fabs(o.x), xb = std::fabs(o.y);
            x[(2 * r) * 256 + n] = xa; x[(2 * r + 1) * 256 + n] = xb;                   // x leaves in natural order
            if (cnc) { prox_cnc_pt(xa, zs[ia], ws[ia], pc); prox_cnc_pt(xb, zs[ib], ws[ib], pc); }
            else     { prox_l1_pt(xa, zs[ia], ws[ia], pc);  prox_l1_pt(xb, zs[ib], ws[ib], pc); }
        }
    }
    for (int row = 0; row < 256; ++row) for (int n = 0; n < 256; ++n) {              // back to natural order (k_state_order<false>)
        z[row * 256 + n] = zs[sl_state_index(row, n)];
        w[row * 256 + n] = ws[sl_state_index(row, n)];
    }
    FILE* o = fopen(argv[2], "wb");
    std::vector<double> d(3 * N);
    for (int i = 0; i < N; ++i) { d[i] = x[i]; d[N + i] = z[i]; d[2 * N + i] = w[i]; }
    fwrite(d.data(), 8, d.size(), o);
    fclose(o);
    return 0;
}

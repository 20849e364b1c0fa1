// Data layout of the fused 256x256 path (shared by kernels_fused256.hip and the g++ host test).
//
// Two real slices a = 2p, b = 2p+1 travel as ONE complex field c = v_a + i v_b ("pair" p), so a
// single complex 2-D FFT transforms both.  k-space blend happens on the Hermitian-symmetrised
// measurements:
//     Yh_s[k] = (m_s[k] y_s[k] + m_s[-k] conj(y_s[-k])) / 2,     Mh_s[k] = (m_s[k] + m_s[-k]) / 2
// needed only on the half plane k2 = 0..128 because the column kernel processes column k2 and its
// mirror 256-k2 together.
//
//   T  : [pair][r][k2]           c32     row-FFT'd field, 512 KiB per pair (in place both ways)
//   Yh : [pair][tile m:9][wave:4][j:16][kl:16][tq:4] float4 {Yh_a.re, Yh_a.im, Yh_b.re, Yh_b.im}
//        for k2 = 16 m + kl (m = 8: k2 = 128 only), k1 = t + 16 j, t = 4 wave + tq
//   Mh : [pair][tile m:9][wave:4][kl:16][tq:4] u64, nibble j = (2 Mh_a) | (2 Mh_b) << 2
// i.e. exactly thread order of the column kernel (lane = kl + 16 tq inside wave `wave` of tile m):
// every wave-level operand load is one contiguous, fully coalesced 1-KiB (Yh) / 512-B (Mh) access,
// and (j, t) is the (register, lane) the value meets after the cooperative FFT of fft16.h.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include "fft16.h"

namespace pnp {

constexpr int F_N = 256;
constexpr int F_HALF = 129;                  // k2 = 0..128

constexpr int F_TILES = 9;                   // 8 tiles of 16 column pairs + the tile of column 128
constexpr size_t YH_PAIR = (size_t)F_TILES * 4 * 16 * 16 * 4;     // float4 per slice pair
constexpr size_t MH_PAIR = (size_t)F_TILES * 4 * 16 * 4;          // u64 per slice pair

PNP_HD size_t yh_index(int pair, int k2, int j, int t) {
    const int m = k2 >> 4, kl = k2 & 15, wv = t >> 2, tq = t & 3;
    return (size_t)pair * YH_PAIR + ((((size_t)(m * 4 + wv) * 16 + j) * 16 + kl) * 4 + tq);
}
PNP_HD size_t mh_index(int pair, int k2, int t) {
    const int m = k2 >> 4, kl = k2 & 15, wv = t >> 2, tq = t & 3;
    return (size_t)pair * MH_PAIR + (((size_t)(m * 4 + wv) * 16 + kl) * 4 + tq);
}

// Hermitian-symmetrised measurement and mask code of one slice at (k1, k2).
// y, mask: the slice's [256][256] arrays.
PNP_HD void hermitian_entry(const c32* y, const uint8_t* mask, int k1, int k2, c32& yh, int& code) {
    const int i1 = k1 * F_N + k2;
    const int i2 = ((F_N - k1) & 255) * F_N + ((F_N - k2) & 255);
    const int m1 = mask[i1] != 0, m2 = mask[i2] != 0;
    const c32 y1 = y[i1], y2 = y[i2];
    // select, do not multiply: an unsampled y entry (possibly NaN/Inf in user data) must not reach the result
    yh = mk(0.5f * ((m1 ? y1.x : 0.0f) + (m2 ? y2.x : 0.0f)), 0.5f * ((m1 ? y1.y : 0.0f) - (m2 ? y2.y : 0.0f)));
    code = m1 + m2;
}

// ----------------------------------------------------------------------------------------------
// "Split chain" tables (k_fcols2 in kernels_fused256.hip; float and double): each thread runs the
// column chain of ONE slice, thread pairs (slice a, slice b) sit in neighbouring lanes:
//   lane = s + 2 kl + 16 tq   (s = slice of the pair, kl = 0..7 column pair in the tile, t = 4 wave + tq)
//   tile m = 0..15: k2 = 8 m + kl (k2 = 0 unused),  tile 16: the self-mirrored columns, kl 0 -> k2 = 0, kl 1 -> k2 = 128
//   Yh2 : [pair][tile 17][wave 4][j 16][lane 64] complex   Yh_s at k1 = t + 16 j
//   Mh2 : [pair][tile 17][wave 4][lane 64] u32, 2 bits per j = 2 Mh_s
// again thread order: every wave-level operand load is one contiguous 512-byte (float) access.
// ----------------------------------------------------------------------------------------------
constexpr int F2_TILES = 17;
constexpr size_t YH2_PAIR = (size_t)F2_TILES * 4 * 16 * 64;      // complex values per slice pair
constexpr size_t MH2_PAIR = (size_t)F2_TILES * 4 * 64;           // u32 per slice pair
PNP_HD int f2_tile(int k2) { return (k2 == 0 || k2 == 128) ? 16 : (k2 >> 3); }
PNP_HD int f2_kl(int k2) { return k2 == 0 ? 0 : (k2 == 128 ? 1 : (k2 & 7)); }
PNP_HD size_t yh2_index(int pair, int k2, int j, int t, int s) {
    const int lane = s + 2 * f2_kl(k2) + 16 * (t & 3);
    return (size_t)pair * YH2_PAIR + ((((size_t)(f2_tile(k2) * 4 + (t >> 2)) * 16 + j) * 64) + lane);
}
PNP_HD size_t mh2_index(int pair, int k2, int t, int s) {
    const int lane = s + 2 * f2_kl(k2) + 16 * (t & 3);
    return (size_t)pair * MH2_PAIR + ((size_t)(f2_tile(k2) * 4 + (t >> 2)) * 64 + lane);
}
// Hermitian-symmetrised measurement and mask code of one slice at (k1, k2), any precision
template <typename R>
PNP_HD void hermitian_entry_t(const cxT<R>* y, const uint8_t* mask, int k1, int k2, cxT<R>& yh, int& code) {
    const int i1 = k1 * F_N + k2;
    const int i2 = ((F_N - k1) & 255) * F_N + ((F_N - k2) & 255);
    const int m1 = mask[i1] != 0, m2 = mask[i2] != 0;
    const cxT<R> y1 = y[i1], y2 = y[i2];
    yh = mk<R>((R)0.5 * ((m1 ? y1.x : (R)0) + (m2 ? y2.x : (R)0)), (R)0.5 * ((m1 ? y1.y : (R)0) - (m2 ? y2.y : (R)0)));
    code = m1 + m2;
}

// ----------------------------------------------------------------------------------------------
// 512 x 512: same scheme, 32 lanes per transform (fft16.h, "512 = 16 points x 32 lanes").
//   T  : [pair][r][phi512(k2)] c32, 2 MiB per pair; phi512: [0]=col 0, [1]=col 256, [2q]=col q, [2q+1]=col 512-q
//   Yh : [pair][tile m:33][wave:4][reg q:16][lane:64] float4,  Mh : [pair][tile][wave][lane] u64
// in the column kernel's thread order: tile m holds column pairs 8m .. 8m+7 (m = 32: column 256),
// lane = kl + 8 tq, t = 8 wave + tq, and after the forward transform lane t, register q hold
// k1 = (t >> 1) + 16 q + 256 (t & 1)  (the k-layout of fft16.h).
// ----------------------------------------------------------------------------------------------
constexpr int F5_N = 512;
constexpr int F5_HALF = 257;                 // k2 = 0..256
constexpr int F5_TILES = 33;
constexpr size_t YH5_PAIR = (size_t)F5_TILES * 4 * 16 * 64;
constexpr size_t MH5_PAIR = (size_t)F5_TILES * 4 * 64;

PNP_HD int phi512(int k) { return k < 256 ? 2 * k : (k == 256 ? 1 : 1025 - 2 * k); }
PNP_HD size_t yh5_index(int pair, int k2, int k1) {
    const int m = k2 >> 3, kl = k2 & 7;
    const int t = 2 * (k1 & 15) + (k1 >> 8), q = (k1 >> 4) & 15;
    const int wv = t >> 3, lane = kl + 8 * (t & 7);
    return (size_t)pair * YH5_PAIR + (((size_t)(m * 4 + wv) * 16 + q) * 64 + lane);
}
PNP_HD size_t mh5_index(int pair, int k2, int t) {
    const int m = k2 >> 3, kl = k2 & 7, wv = t >> 3, lane = kl + 8 * (t & 7);
    return (size_t)pair * MH5_PAIR + ((size_t)(m * 4 + wv) * 64 + lane);
}
PNP_HD void hermitian_entry512(const c32* y, const uint8_t* mask, int k1, int k2, c32& yh, int& code) {
    const int i1 = k1 * F5_N + k2;
    const int i2 = ((F5_N - k1) & 511) * F5_N + ((F5_N - k2) & 511);
    const int m1 = mask[i1] != 0, m2 = mask[i2] != 0;
    const c32 y1 = y[i1], y2 = y[i2];
    // select, do not multiply: an unsampled y entry (possibly NaN/Inf in user data) must not reach the result
    yh = mk(0.5f * ((m1 ? y1.x : 0.0f) + (m2 ? y2.x : 0.0f)), 0.5f * ((m1 ? y1.y : 0.0f) - (m2 ? y2.y : 0.0f)));
    code = m1 + m2;
}

}  // namespace pnp

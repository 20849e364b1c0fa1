// Index maps of the slice-resident 256x256 kernel (kernels_slice256.hip), shared with the g++ host
// emulation (tests/host/slice_resident_emulation.cpp).
//
// One 512-thread workgroup (8 waves, 2 per SIMD, 256 VGPRs each) keeps ONE real slice on a compute
// unit for a whole ADMM run: the 65536 values live in the register file (128 VGPRs per thread, four
// "register sets" of 16 complex values) as 128 complex rows or 128 complex columns of 256, and only
// z, w (and the Hermitian measurement table) travel to HBM.
//
//   row form   : row pair r = 0..127 carries c_r[n] = v[2r][n] + i v[2r+1][n]; its transform C_r[k]
//                is held by the 16 lanes of a group as C_r[t + 16 j] (lane t, register j).
//                thread (wave wv, lane l), register set s:  r = 32 s + 4 wv + (l >> 4),  t = l & 15
//   column form: column c = 0..127; c >= 1 is k-space column k2 = c of the real slice's row
//                transforms V_rho[k2] (rho = 0..255), c = 0 packs the two real columns k2 = 0 and
//                k2 = 128 as V_rho[0] + i V_rho[128].  Same thread shape:
//                c = 32 s + 4 wv + (l >> 4),  lane t holds rho (or k1) = t + 16 j
//
// Row form <-> column form goes through LDS in two passes (the buffer holds half the field):
//   pass p moves the columns c = 64 p .. 64 p + 63 (register sets 2p and 2p + 1 of the column form).  A row pair needs, per column c, C_r[c] and its
//   mirror C_r[256 - c] (for c = 0: C_r[0] and C_r[128]) because
//     V_2r[k2] = (C_r[k2] + conj C_r[-k2]) / 2,   V_2r+1[k2] = (C_r[k2] - conj C_r[-k2]) / (2i)
//   buffer element (r, slot): slot = c - 64 p for the direct value, SL_M + c - 64 p for the mirror.
// The way back is the exact mirror: the thread pair (rho = 2r, 2r+1) of column c forms
//   D_r[c] = U_2r + i U_2r+1 (direct slot) and D_r[256 - c] = conj U_2r + i conj U_2r+1 (mirror slot).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include "fft16.h"

namespace pnp {

constexpr int SL_P = 132;     // pitch (complex) of a row pair in the transposition buffer; 132 % 32 == 4:
                              // 8 consecutive row pairs x 2 neighbouring columns hit 16 distinct 8-byte banks
constexpr int SL_M = 66;      // mirrored half starts here; 66 % 4 == 2 keeps direct / mirror stores of a lane pair apart
constexpr int SL_BUF = 128 * SL_P;                       // complex elements of the transposition buffer

// pass in which k-space column k (0..255) of a row pair crosses, and its slot
PNP_HD int sl_pass(int k) { return (k < 64 || k > 192 || k == 128) ? 0 : 1; }
PNP_HD int sl_slot(int k) {
    if (k < 64) return k;                      // pass 0, direct:  c = k
    if (k == 128) return SL_M;                 // pass 0, partner of the packed column c = 0
    if (k > 192) return SL_M + 256 - k;        // pass 0, mirror of c = 256 - k = 1..63
    if (k < 128) return k - 64;                // pass 1, direct:  c = k = 64..127
    return SL_M + 192 - k;                     // pass 1, mirror of c = 256 - k = 64..127  (k = 129..192)
}
// thread shape shared by both forms
constexpr int SL_WAVES = 8, SL_SETS = 4;
PNP_HD int sl_unit(int set, int wv, int lane) { return 32 * set + 4 * wv + (lane >> 4); }     // r or c

// per-slice operand tables in column-form thread order
//   Yh3 : [slice][set 4][wave 8][j/2 8][lane 64][j%2] complex   Yh at (k1 = t + 16 j, k2 = c); for c = 0: k2 = 0.
//         A wave's operands of a set are ONE contiguous 8 KiB block read by eight 16-byte-per-lane accesses (two j each)
//   Mh3 : [slice][set 4][wave 8][lane 64] u32, 2 bits per j = 2 Mh
//   Ys3 : [slice][256] complex, Ms3 : [slice][16] u32 -- the same for k2 = 128 (second half of c = 0), lane t, bits j
constexpr size_t YH3_SLICE = 4 * 16 * 8 * 64;
constexpr size_t MH3_SLICE = 4 * 8 * 64;
PNP_HD size_t yh3_index(int slice, int set, int j, int wv, int lane) {
    return (size_t)slice * YH3_SLICE + ((((size_t)set * 8 + wv) * 8 + (j >> 1)) * 64 + lane) * 2 + (j & 1);
}
PNP_HD size_t mh3_index(int slice, int set, int wv, int lane) { return (size_t)slice * MH3_SLICE + (((size_t)set * 8 + wv) * 64 + lane); }

// State arrays (z, w) of a slice-resident run live in HBM in "slice order": rows stay where they are, but the 512 values
// of a ROW PAIR r (image rows 2r, 2r + 1 = re / im of the packed complex row) are stored in the thread order of the row
// transform, so that the z-/w-update works on the transform's own registers (no natural-order staging through LDS) and
// every access is still a full 16-byte lane access of a 256-byte contiguous run per 16-lane group:
//   pixel n = t + 16 j of row 2r + sel  ->  float index  512 r + 64 (j >> 1) + 4 t + 2 (j & 1) + sel
// i.e. lane t's q-th 16-byte access (q = 0..7) holds (row 2r, row 2r+1) x (j = 2q, 2q + 1) = its registers a[2q], a[2q + 1].
// pnp_get_state / pnp_set_state / the other kernel families see the natural [256][256] order: api.hip converts in place
// (k_state_order) when a run switches families.  x is always natural.
PNP_HD size_t sl_state_index(int row, int n) {                 // within the slice's 65536 floats
    return (size_t)(row >> 1) * 512 + 64 * (n >> 5) + 4 * (n & 15) + 2 * ((n >> 4) & 1) + (row & 1);
}
// (A variant that interleaves the four row pairs of a wave's register set so that every wave access is 1 KiB contiguous
// instead of 4 x 256 bytes measured the same, 10.3 k it/s: the kernel is at the memory system's limit for its bytes.)

// the packed column c = 0 after its transform: G[k1] = A[k1] + i B[k1] with A, B the transforms of the
// REAL columns k2 = 0 and k2 = 128:  A = unpack_a(G[k1], G[-k1]),  B = unpack_b(G[k1], G[-k1]),
// and back  G' = repack_p(A', B') = A' + i B'   (fft16.h).

}  // namespace pnp

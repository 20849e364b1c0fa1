"""Build-time guards for the hand-tuned gfx950 kernels (CPU only: hipcc cross-compiles without a GPU).

The slice-resident kernel sits on a resource cliff: 256 VGPRs x 512 threads is the whole
register file of a compute unit, a spill or a second code instance halves the headline, and a
`buffer_store_dwordx4` with an SGPR soffset directly followed by a VALU write to its data registers stores the
NEW value on gfx950 (hipcc only protects the form with a constant soffset).  A compiler bump or an innocent
edit can bring any of these back silently, so this file compiles the kernels to gfx950 assembly
(`hipcc --cuda-device-only -S`, the Makefile's flags) and asserts what the measurements depend on:

  * k_slice<1|2|3>: 0 bytes of scratch, <= 256 VGPRs, 2 waves per SIMD, LDS <= 160 KiB, code size bounded;
  * k5_cols (512x512 column kernel): <= 168 VGPRs (3 waves per SIMD stay possible), 0 scratch;
  * every kernel of the library: no >= 96-bit buffer store with an SGPR soffset whose data registers are
    written by a VALU instruction inside the hazard window.
"""
import os
import re
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, 'pnp_admm_cnc_mri_amd', 'csrc')
HIPCC = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
SOURCES = ['kernels_slice256.hip', 'kernels_fused256.hip', 'kernels_fused512.hip', 'kernels_generic.hip', 'kernels_conv.hip', 'kernels_conv_f16x3.hip',
           'kernels_conv_f16x3_wide.hip', 'kernels_pix2x2_f16x3.hip', 'api.hip']
FLAGS = ['-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-ffp-contract=off']     # = csrc/Makefile CXXFLAGS

import importlib.util
_spec = importlib.util.spec_from_file_location('isa_scan', os.path.join(CSRC, 'tools', 'isa_scan.py'))
isa_scan = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(isa_scan)          # the scanner the Makefile runs on every build (`make check`)
kernels_of, store_hazards, dma_order_violations = isa_scan.kernels_of, isa_scan.store_hazards, isa_scan.dma_order_violations

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason='hipcc not installed')


def _compile(args):
    src, out = args
    extra = ['-fno-slp-vectorize'] if src in ('kernels_conv_f16x3.hip', 'kernels_conv_f16x3_wide.hip', 'kernels_pix2x2_f16x3.hip') else []          # = the Makefile's per-file flag
    r = subprocess.run([HIPCC] + FLAGS + extra + ['--cuda-device-only', '-S', os.path.join(CSRC, src), '-o', out],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=CSRC)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    return out


@pytest.fixture(scope='module')
def asm(tmp_path_factory):
    d = tmp_path_factory.mktemp('gfx950_asm')
    jobs = [(s, str(d / (s[:-4] + '.s'))) for s in SOURCES]
    with ThreadPoolExecutor(3) as ex:
        outs = list(ex.map(_compile, jobs))
    return {s: open(o).read() for s, o in zip(SOURCES, outs)}



def test_slice_resident_kernel_stays_off_the_cliff(asm):
    ks = {n: k for n, k in kernels_of(asm['kernels_slice256.hip']).items() if 'k_slice' in n}
    assert len(ks) == 3, sorted(ks)                                  # PROX = 1, 2, 3
    for name, k in ks.items():
        i = k['info']
        assert i['ScratchSize'] == 0, (name, i)                      # a spill costs 10-30 % ("how it got there")
        assert i['NumVgprs'] + i['NumAgprs'] <= 256, (name, i)
        assert i['Occupancy'] == 2, (name, i)                        # 8 waves x 256 VGPRs = one workgroup per compute unit
        assert i['LDSByteSize'] <= 160 * 1024, (name, i)
        # the loop body does not fit the 64 KiB instruction cache either way (measured with 84-92 KB); what must not
        # come back is the 116 KB of two row-phase instances
        assert i['codeLenInByte'] <= 96 * 1024, (name, i)


def test_slice_resident_kernel_instruction_budget(asm):
    """What round 3's two passes over the compiled ISA removed must not come back unnoticed with a compiler bump or an edit
    (DESIGN.md 4.1): the conservative hazard `s_nop`s behind inline asm (294 per wave-iteration before the packed complex
    product became one asm statement), hipcc's scalarised forms of the pointwise update (four `v_sub_f32` per access, `v_and`
    + packed add for |a| + w), per-access `v_add_u32` for LDS offsets beyond the 16-bit immediate, the packed column's selects
    in all eight waves.  Budgets = the whole kernel (prologue + loop + the last iteration's extras), 8-10 % above today's
    counts (CNC: 6478 vector instructions of which 4600 packed, 191 s_nop, 48 v_sub_f32, 165 v_cndmask)."""
    ks = {n: k for n, k in kernels_of(asm['kernels_slice256.hip']).items() if 'k_slice' in n}
    cnc = [k for n, k in ks.items() if 'ILi2E' in n]
    assert len(cnc) == 1, sorted(ks)
    body = cnc[0]['body']
    valu = [i for i in body if i.startswith('v_')]
    count = lambda pre: sum(1 for i in body if i.startswith(pre))
    stats = {'valu': len(valu), 'packed': count('v_pk_'), 's_nop': count('s_nop'), 'v_sub_f32': count('v_sub_f32'),
             'v_cndmask': count('v_cndmask'), 'v_mov_b32': count('v_mov_b32'), 'ds': count('ds_'), 'scratch': count('scratch_')}
    assert stats['scratch'] == 0, stats
    assert stats['valu'] <= 7000, stats
    assert stats['s_nop'] <= 260, stats
    assert stats['v_sub_f32'] <= 120, stats                          # 342 with the scalarised v = z - w
    assert stats['v_cndmask'] <= 230, stats
    assert stats['packed'] >= 0.62 * stats['valu'], stats            # the transforms stay on packed fp32


def test_512_column_kernel_register_budget(asm):
    ks = {n: k for n, k in kernels_of(asm['kernels_fused512.hip']).items() if 'k5_cols' in n}
    assert ks
    for name, k in ks.items():
        assert k['info']['ScratchSize'] == 0 and k['info']['NumVgprs'] <= 168, (name, k['info'])





def test_conv_kernel_keeps_two_workgroups_per_compute_unit(asm):
    """k_conv3x3_c64 counts on TWO persistent workgroups per compute unit -- one's staging and epilogue under
    the other's MFMAs: no scratch, at most 256 VGPRs + AGPRs (2 waves per SIMD), at most 80 KiB of LDS; its MFMAs are the fp32
    32x32x2 form; the generic blend kernel stays at three workgroups per unit (<= 168 VGPRs, 0 scratch)."""
    ks = kernels_of(asm['kernels_conv.hip'])
    body = {n: k for n, k in ks.items() if 'k_conv3x3_c64' in n}
    assert len(body) == 4, sorted(ks)                                # dilations 1..4
    for n, k in body.items():
        i = k['info']
        dil = int(n.split('k_conv3x3_c64ILi')[1][0])
        two = dil <= 2                                                # dilation 1, 2: two workgroups per compute unit; 3, 4: one (82 / 102 KiB of LDS)
        assert i['ScratchSize'] == 0 and i['LDSByteSize'] <= (80 if two else 160) * 1024, (n, i)
        assert i['NumVgprs'] + i['NumAgprs'] <= (256 if two else 512), (n, i)
        mf = [x for x in k['body'] if x.startswith('v_mfma')]
        assert len(mf) == 64 and all(x.startswith('v_mfma_f32_32x32x2_f32') for x in mf), (n, len(mf))   # 8 groups x 4 steps x 2 accumulators, taps looped
    for n, k in ks.items():
        assert k['info']['ScratchSize'] == 0, (n, k['info'])
    blend = [k for n, k in kernels_of(asm['kernels_generic.hip']).items() if 'k_cols16ILb1ELi1ELb1E' in n]
    assert len(blend) == 1 and blend[0]['info']['NumVgprs'] <= 168 and blend[0]['info']['ScratchSize'] == 0, blend[0]['info']


def test_f16x3_conv_kernel_resources(asm):
    """k_conv3x3_c64_h3: at dilation 1 two workgroups share a compute unit -- 81 728 bytes of LDS each (input tile +
    two weight buffers: the 160 KiB exactly), at most 256 registers, no scratch; every matrix instruction is the half form with
    float32 accumulation, 48 per tap (2 K steps of 32 x 2 M tiles x 4 N tiles x 3 products; the nine taps are unrolled), and the
    weights reach LDS by LDS-DMA (4 per tap and wave + the first tap's)."""
    ks = {n: k for n, k in kernels_of(asm['kernels_conv_f16x3.hip']).items() if 'k_conv3x3_c64_h3' in n}
    assert len(ks) == 4, sorted(ks)
    for n, k in ks.items():
        i = k['info']
        dil = int(n.split('k_conv3x3_c64_h3ILi')[1][0])
        assert i['ScratchSize'] == 0, (n, i)
        assert i['LDSByteSize'] <= (80 if dil == 1 else 160) * 1024, (n, i)
        assert i['NumVgprs'] + i['NumAgprs'] <= (256 if dil == 1 else 512), (n, i)
        mf = [x for x in k['body'] if x.startswith('v_mfma')]
        # two instances of the chunk code (the first chunk of an item starts its accumulators from a constant-zero C operand)
        assert len(mf) == 2 * 9 * 48 and all(x.startswith('v_mfma_f32_16x16x32_f16') for x in mf), (n, len(mf), mf[:2])
        assert sum(1 for x in mf if x.split(';')[0].rstrip().endswith(', 0')) == 16, n
        assert sum(1 for x in k['body'] if x.startswith('buffer_load_dwordx4') and x.split(';')[0].rstrip().endswith(' lds')) == 4 + 2 * 36, n


def test_f16x3_weight_dma_is_older_than_the_loads_counted_behind_it(asm):
    """The per-tap `s_waitcnt vmcnt(k)` of k_conv3x3_c64_h3 leaves the k loads of an input-prefetch piece in flight and is right only
    while the tap's four LDS-DMAs were issued BEFORE them (in-order completion); the source pins that with a sched_barrier and this
    walk over the compiled ISA proves it for every dilation: no group of DMAs is issued before the group before it was published by
    a barrier with no DMA possibly in flight, and no wave ends with one in flight.  The scanner itself on the shapes it must tell apart:"""
    D, L = 'buffer_load_dwordx4 v1, s[0:3], s5 offen lds', 'buffer_load_dwordx4 v[2:5], v1, s[0:3], 0 offen'
    tail = [D] * 4 + ['s_waitcnt vmcnt(0)', 's_barrier', 's_endpgm']
    assert not dma_order_violations([D] * 4 + [L, L, 's_waitcnt vmcnt(2)', 's_barrier'] + tail)
    assert not dma_order_violations([L] * 5 + [D, D, L, D, D, 's_waitcnt vmcnt(0)', 's_barrier'] + tail)       # a prologue may interleave: it drains
    assert dma_order_violations([D, D, D, L, D, L, 's_waitcnt vmcnt(2)', 's_barrier'] + tail)                  # a load moved ahead of a DMA
    assert dma_order_violations([D] * 4 + [L, L, 's_waitcnt vmcnt(3)', 's_barrier'] + tail)                    # a count one too high
    assert dma_order_violations([D] * 4 + ['s_waitcnt vmcnt(0)', 's_barrier'] + [D] * 4 + ['s_endpgm'])        # ends with a DMA in flight
    ks = {n: k for n, k in kernels_of(asm['kernels_conv_f16x3.hip']).items() if 'k_conv3x3_c64_h3' in n}
    assert len(ks) == 4
    for n, k in ks.items():
        assert sum(1 for x in k['body'] if x.startswith('buffer_load') and x.rstrip().endswith(' lds')) >= 76, n
        assert not dma_order_violations(k['body'], labels=k['labels']), (n, dma_order_violations(k['body'], labels=k['labels'])[:3])
    # round 6 (advisor): back edges -- the hand-over from a loop's last group to its first is walked too
    loop = ['s_barrier'] + [D] * 4 + [L, L, 's_waitcnt vmcnt(2)', 's_cbranch_scc1 .LBB0_1']
    assert not dma_order_violations([D] * 4 + ['s_waitcnt vmcnt(0)'] + loop + ['s_waitcnt vmcnt(0)', 's_endpgm'], labels={'.LBB0_1': 5})
    bad_loop = ['s_barrier'] + [D] * 4 + [L, L, 's_waitcnt vmcnt(6)', 's_cbranch_scc1 .LBB0_1']      # the wait leaves the group in flight ...
    assert not dma_order_violations([D] * 4 + ['s_waitcnt vmcnt(0)'] + bad_loop + ['s_waitcnt vmcnt(0)', 's_endpgm'])     # ... a linear walk misses it,
    assert dma_order_violations([D] * 4 + ['s_waitcnt vmcnt(0)'] + bad_loop + ['s_waitcnt vmcnt(0)', 's_endpgm'], labels={'.LBB0_1': 5})   # the second pass does not
    # ahead = 1 (the wide kernel: three buffers, the group requested in tap T is waited for at the end of tap T + 1)
    wide = [D] * 8 + ['s_waitcnt vmcnt(0)'] + ['s_barrier'] + [D] * 4 + [L, 's_waitcnt vmcnt(5)', 's_cbranch_scc1 .LBB0_1'] + ['s_waitcnt vmcnt(0)', 's_endpgm']
    assert not dma_order_violations(wide, ahead=1, labels={'.LBB0_1': 9})
    assert dma_order_violations(wide, ahead=0, labels={'.LBB0_1': 9})                                   # the narrow kernel's rule forbids it
    assert dma_order_violations([x.replace('vmcnt(5)', 'vmcnt(9)') for x in wide], ahead=1, labels={'.LBB0_1': 9})      # two groups in flight at a barrier


def test_wide_f16x3_kernel_resources(asm):
    """k_conv3x3_h3w (csrc/kernels_conv_f16x3_wide.hip): ONE workgroup of eight waves per compute unit (<= 256 registers, no scratch, 150 KiB
    of LDS), 2 x 9 x 96 half-precision matrix instructions (two instances of the chunk code), inside the compute waves' tap loop nothing but
    LDS reads, MFMAs, waits and barriers -- the point of the kernel --, and a weight stream (two waves of their own, three buffers) that the compiled
    ISA waits for before every barrier, back edges included."""
    ks = {n: k for n, k in kernels_of(asm['kernels_conv_f16x3_wide.hip']).items() if 'k_conv3x3_h3w' in n}
    assert len(ks) == 1, sorted(ks)
    for n, k in ks.items():
        i = k['info']
        assert i['ScratchSize'] == 0 and i['NumVgprs'] + i['NumAgprs'] <= 256 and i['LDSByteSize'] == 153920, (n, i)     # 88 128 (tile) + 4 x 16 384 (weights) + 256 (biases)
        mf = [j for j, x in enumerate(k['body']) if x.startswith('v_mfma')]
        assert len(mf) == 2 * 9 * 96 and all(k['body'][j].startswith('v_mfma_f32_16x16x32_f16') for j in mf), (n, len(mf))
        assert sum(1 for j in mf if k['body'][j].split(';')[0].rstrip().endswith(', 0')) == 32, n        # 16 main + 16 correction accumulators start from 0
        # between the first and the last MFMA of each chunk instance: ds_read_b128, MFMA, s_waitcnt, s_barrier (+ s_nop) -- and the handful of
        # scalar / vector adds that rotate the weight buffer (four buffers, nine taps per chunk)
        for lo, hi in ((mf[0], mf[863]), (mf[864], mf[-1])):
            other = [x for x in k['body'][lo:hi] if not x.startswith(('v_mfma', 'ds_read_b128', 's_waitcnt', 's_barrier', 's_nop'))]
            assert len(other) <= 16 and all(x.startswith(('v_add_u32', 'v_lshl_add_u32', 's_add', 's_and', 's_xor', 's_lshl')) for x in other), (n, other[:8])
            assert sum(1 for x in k['body'][lo:hi] if x.startswith('ds_read_b128')) >= 9 * 32 - 16, n           # (tap 0's first twelve reads and its second weight pair stand in front of the first MFMA)
        # four helper waves x four DMAs per tap: three taps requested (and drained) in the prologue, nine in the loop, each group waited for one tap later
        assert sum(1 for x in k['body'] if x.startswith('buffer_load') and x.rstrip().endswith(' lds')) == 12 + 36, n
        v = dma_order_violations(k['body'], ahead=1, primed=1, labels=k['labels'])
        assert not v, (n, v[:3])
        assert dma_order_violations(k['body'], ahead=0, primed=2, labels=k['labels'])                # (the narrow kernel's rule: the scan does tell the protocols apart)


def test_pix2x2_kernel_resources(asm):
    """k_pix2x2_h3 (DRUNet's 2 x 2 strided / transposed convolutions, csrc/kernels_pix2x2_f16x3.hip): four instances (down / up, with / without
    the added second input), two workgroups per compute unit (<= 80 KiB of LDS, <= 256 registers, no scratch -- the variant with a
    second input sits close to the line), 48 half-precision matrix instructions per chunk in two unrolled chunks, the weights by LDS-DMA
    in an order the counted waits can rely on."""
    ks = {n: k for n, k in kernels_of(asm['kernels_pix2x2_f16x3.hip']).items() if 'k_pix2x2_h3' in n}
    assert len(ks) == 4, sorted(ks)
    for n, k in ks.items():
        i = k['info']
        assert i['ScratchSize'] == 0 and i['NumVgprs'] + i['NumAgprs'] <= 256 and i['LDSByteSize'] <= 80 * 1024, (n, i)
        mf = [x for x in k['body'] if x.startswith('v_mfma')]
        assert len(mf) == 96 and all(x.startswith('v_mfma_f32_16x16x32_f16') for x in mf), (n, len(mf))
        assert sum(1 for x in k['body'] if x.startswith('buffer_load') and x.rstrip().endswith(' lds')) == 12, n
        assert not dma_order_violations(k['body'], labels=k['labels']), (n, dma_order_violations(k['body'], labels=k['labels'])[:3])


def test_hazard_scanner_sees_the_pattern_that_bit_us():
    """the sequence of DESIGN.md section 4.1 ("Buffer-store hazard") and its harmless neighbours"""
    assert store_hazards(['buffer_store_dwordx4 v[28:31], v224, s[48:51], s83 offen', 'v_mov_b32_e32 v28, v5'])
    assert store_hazards(['buffer_store_dwordx4 v[28:31], v224, s[48:51], s83 offen', 's_add_u32 s1, s2, s3',
                          'v_pk_add_f32 v[30:31], v[2:3], v[4:5]'])
    assert not store_hazards(['buffer_store_dwordx4 v[28:31], v224, s[48:51], 0 offen', 'v_mov_b32_e32 v28, v5'])
    assert not store_hazards(['buffer_store_dwordx4 v[28:31], v224, s[48:51], s83 offen', 's_nop 1', 'v_mov_b32_e32 v28, v5'])
    assert not store_hazards(['buffer_store_dwordx4 v[28:31], v224, s[48:51], s83 offen', 'v_mov_b32_e32 v32, v5',
                              'v_cmp_lt_f32_e32 vcc, v28, v29'])
    assert not store_hazards(['buffer_store_dwordx2 v[28:29], v224, s[48:51], s83 offen', 'v_mov_b32_e32 v28, v5'])


def test_no_wide_buffer_store_is_overwritten_in_its_hazard_window(asm):
    n_stores = 0
    for src, text in asm.items():
        for name, k in kernels_of(text).items():
            n_stores += sum(1 for ins in k['body'] if ins.startswith('buffer_store_dwordx4'))
            bad = store_hazards(k['body'])
            assert not bad, '%s: %s: %s' % (src, name, bad[:3])
    assert n_stores >= 100                                           # the slice kernel's state stores were really scanned

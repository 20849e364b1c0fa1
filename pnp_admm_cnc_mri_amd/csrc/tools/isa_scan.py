#!/usr/bin/env python3
"""ISA scan of the gfx950 assembly of libpnpmri.so's kernels -- part of the BUILD (`make` runs it on every translation unit,
`make check` alone), and the library behind tests/test_build_guard.py.

What it guards ("Buffer-store hazard"): on gfx950 a `buffer_store_dwordx3/x4` reads its data
registers over several cycles, and hipcc inserts the wait states a following VALU write to those registers needs only
when the store's soffset is NOT an SGPR.  `buffer_store_dwordx4 v[28:31], v224, s[48:51], s83 offen` directly followed by
`v_mov_b32 v28, ...` stored the NEW v28 for the last four lanes of every 16 -- sporadic wrong z values.  The kernels
therefore fold the uniform offset into voffset (st4() in kernels_slice256.hip); this scan fails the build if an edit or a
toolchain bump brings the unprotected form back anywhere in the library.

usage: isa_scan.py file.s [file.s ...]      exit status 1 and one line per finding when a hazard is present
"""
import re
import sys


def kernels_of(text):
    """{mangled kernel name: {'body': [instruction lines], 'info': {key: int}}} of one assembly file."""
    res = {}
    lines = text.splitlines()
    i = 0
    while i < len(lines):
        m = re.match(r'^(_Z\w+):\s*(;.*)?$', lines[i])
        if not m:
            i += 1
            continue
        name, body = m.group(1), []
        i += 1
        while i < len(lines) and not lines[i].startswith('.Lfunc_end'):
            ln = lines[i].strip()
            if ln and not ln.startswith((';', '.')) and not ln.endswith(':'):
                body.append(ln.split(';')[0].strip())
            i += 1
        info = {}
        while i < len(lines) and not re.match(r'^_Z\w+:', lines[i]):
            mm = re.match(r'^; (codeLenInByte|NumVgprs|NumAgprs|ScratchSize|Occupancy|LDSByteSize|TotalNumSgprs)\s*[:=]\s*(\d+)', lines[i])
            if mm:
                info[mm.group(1)] = int(mm.group(2))
            if lines[i].startswith('; COMPUTE_PGM_RSRC2:TGID_Z_EN'):
                break
            i += 1
        if 'codeLenInByte' in info:
            res[name] = {'body': body, 'info': info}
    return res


def _regs(tok):
    """VGPR numbers named by an operand token: 'v12' -> {12}, 'v[4:7]' -> {4,5,6,7}, anything else -> {}"""
    m = re.fullmatch(r'v(\d+)', tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def _valu_dest(ins):
    """VGPRs written by a VALU instruction (first operand of v_* except compares / readlanes, which write SGPRs)."""
    op, _, rest = ins.partition(' ')
    if not op.startswith('v_') or op.startswith(('v_cmp', 'v_readlane', 'v_readfirstlane', 'v_nop')):
        return set()
    return _regs(rest.split(',')[0].strip())


def store_hazards(body, window=2):
    """buffer_store_dwordx3/x4 with an SGPR soffset followed, within `window` wait states, by a VALU write to the
    store's data registers.  Every instruction counts as one wait state, s_nop N as N + 1."""
    bad = []
    for i, ins in enumerate(body):
        m = re.match(r'buffer_store_dwordx[34]\s+(v\[\d+:\d+\]),\s*([^,]+),\s*(s\[\d+:\d+\]),\s*(\S+)', ins)
        if not m:
            continue
        soffset = m.group(4).rstrip(',')
        if not re.fullmatch(r's\d+|m0|ttmp\d+', soffset):           # constant soffset: hipcc inserts the wait states itself
            continue
        data, ws, j = _regs(m.group(1)), 0, i + 1
        while j < len(body) and ws < window:
            nxt = body[j]
            if _valu_dest(nxt) & data:
                bad.append((ins, nxt))
                break
            mm = re.match(r's_nop\s+(\d+)', nxt)
            ws += int(mm.group(1)) + 1 if mm else 1
            j += 1
    return bad


def dma_order_violations(body, group=4):
    """LDS-DMA ordering of the f16x3 convolution kernels (kernels_conv_f16x3.hip): a tap's weights are written into LDS by
    `buffer_load_dwordx4 ... lds` (D below) and published to the other waves by `s_waitcnt vmcnt(n)` + `s_barrier`, where n counts the
    plain loads / stores (L) issued BEHIND the DMAs that may stay in flight.  That count is right only while every D of a group is
    older than those L: vector-memory operations complete in order, so `vmcnt(n)` retires everything but the youngest n.  The walk
    keeps the operations that may still be in flight (a wait truncates the list to its youngest n), calls a barrier CLEAN when no D is
    among them, and reports (a) a new group of D issued although no clean barrier followed the group before it -- a buffer would be
    overwritten, or read, with its DMA unfinished -- and (b) a kernel that ends (s_endpgm) with a D possibly in flight.  A plain access
    that slips between or in front of a group's DMAs with a counted wait behind it leaves a D among the youngest n: the barrier is not
    clean and (a) fires at the next group.  The walk is over the linear instruction order (loops are not followed)."""
    bad, inflight, dirty, run = [], [], False, 0
    for idx, ins in enumerate(body):
        if ins.startswith(('buffer_load', 'buffer_store', 'global_load', 'global_store', 'flat_load', 'flat_store', 'buffer_atomic', 'global_atomic')):
            d = ins.startswith('buffer_load') and ins.rstrip().endswith(' lds')
            if d:
                if run % group == 0 and dirty:
                    bad.append((idx, 'LDS-DMA group issued before the group before it was published by a clean barrier', ins))
                run += 1
                if run % group == 0:
                    dirty = True
            inflight.append('D' if d else 'L')
            continue
        m = re.match(r's_waitcnt\b.*vmcnt\((\d+)\)', ins)
        if m:
            n = int(m.group(1))
            inflight = inflight[len(inflight) - n:] if n else []
        elif ins.startswith('s_barrier'):
            if 'D' not in inflight:
                dirty = False
        elif ins.startswith('s_endpgm') and 'D' in inflight:
            bad.append((idx, 'wave may end with an LDS-DMA in flight', ins))
    return bad


def scan_text(text):
    """[(kernel, store, overwriting instruction)] of one assembly file, and the number of wide buffer stores seen."""
    bad, n = [], 0
    for name, k in kernels_of(text).items():
        n += sum(1 for ins in k['body'] if re.match(r'buffer_store_dwordx[34]', ins))
        bad += [(name, a, b) for a, b in store_hazards(k['body'])]
        if '_h3' in name and any(ins.startswith('buffer_load') and ins.rstrip().endswith(' lds') for ins in k['body']):
            bad += [(name, 'LDS-DMA order: ' + why, ins) for _, why, ins in dma_order_violations(k['body'])]
    return bad, n


def main(paths):
    total, rc = 0, 0
    for p in paths:
        bad, n = scan_text(open(p).read())
        total += n
        for name, a, b in bad:
            sys.stderr.write('%s: %s: %s:\n    %s\n    %s\n' % (p, name, 'hazard' if a.startswith('LDS-DMA') else 'wide buffer store with an SGPR soffset overwritten inside its hazard window', a, b))
            rc = 1
    sys.stderr.write('isa_scan: %d wide buffer stores in %d file(s): %s\n' % (total, len(paths), 'HAZARD' if rc else 'ok'))
    return rc


if __name__ == '__main__':
    sys.exit(main(sys.argv[1:]))

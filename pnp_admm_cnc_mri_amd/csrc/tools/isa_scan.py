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
        name, body, labels = m.group(1), [], {}
        i += 1
        while i < len(lines) and not lines[i].startswith('.Lfunc_end'):
            ln = lines[i].strip()
            ml = re.match(r'^(\.LBB\w+):', ln)
            if ml:
                labels[ml.group(1)] = len(body)                  # index of the first instruction behind the label
            elif ln and not ln.startswith((';', '.')) and not ln.endswith(':'):
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
            res[name] = {'body': body, 'info': info, 'labels': labels}
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


def dma_order_violations(body, group=4, ahead=0, labels=None, primed=0):
    """LDS-DMA ordering of the f16x3 convolution kernels (kernels_conv_f16x3.hip / kernels_conv_f16x3_wide.hip): a tap's weights
    are written into LDS by `buffer_load_dwordx4 ... lds` (D below) and published to the other waves by `s_waitcnt vmcnt(n)` + `s_barrier`,
    where n counts the operations issued BEHIND the DMAs that may stay in flight.  That count is right only while every D of a group
    is older than those operations: vector-memory operations complete in order, so `vmcnt(n)` retires everything but the youngest n.
    The walk keeps the operations that may still be in flight (a wait truncates the list to its youngest n); at a barrier the groups
    whose D are all retired count as PUBLISHED.  `ahead` = how many groups may be in flight across a barrier: 0 for the narrow kernel
    (double-buffered: a tap's group lands before the next barrier), 1 for the wide one (triple-buffered: the group requested in tap T is
    waited for at the end of tap T + 1).  `primed`: groups more than `ahead` that a prologue may request in front of the first barrier
    (it drains them all before it arrives there; the wide kernel fills three of its four buffers this way).  Reported: (a) group g issued while fewer than g - ahead groups are published -- a buffer would
    be overwritten, or read, with its DMA unfinished; (b) a kernel that ends (s_endpgm) with a D possibly in flight.  A plain access that
    slips between or in front of a group's DMAs with a counted wait behind it leaves a D among the youngest n: that group stays
    unpublished and (a) fires.  `labels` ({label: index into body}, kernels_of): every BACKWARD branch is followed again (ahead + 1 more times) -- the loop
    body is walked with the state the pass before left, so the hand-over from a loop's last group to its first is checked
    too (round 5's walk was linear and never saw a back edge)."""
    bad = []
    st = {'inflight': [], 'run': 0, 'published': primed}

    def step(idx, ins):
        if ins.startswith(('buffer_load', 'buffer_store', 'global_load', 'global_store', 'flat_load', 'flat_store', 'buffer_atomic', 'global_atomic',
                           'scratch_load', 'scratch_store')):
            d = ins.startswith('buffer_load') and ins.rstrip().endswith(' lds')
            if d:
                if st['run'] % group == 0 and st['published'] < st['run'] // group - ahead:
                    bad.append((idx, 'LDS-DMA group issued before the group %s it was published by a barrier' % ('before' if ahead == 0 else '%d before' % (ahead + 1)), ins))
                st['run'] += 1
            st['inflight'].append('D' if d else 'L')
            return
        m = re.match(r's_waitcnt\b.*vmcnt\((\d+)\)', ins)
        if m:
            n = int(m.group(1))
            st['inflight'] = st['inflight'][len(st['inflight']) - n:] if n else []
        elif ins.startswith('s_barrier'):
            nd = st['inflight'].count('D')
            st['published'] = max(st['published'], st['run'] // group - (nd + group - 1) // group)
        elif ins.startswith('s_endpgm') and 'D' in st['inflight']:
            bad.append((idx, 'wave may end with an LDS-DMA in flight', ins))

    for idx, ins in enumerate(body):
        step(idx, ins)
        mb = re.match(r's_c?branch\w*\s+(\.LBB\w+)', ins) if labels else None
        if mb and labels.get(mb.group(1), idx + 1) <= idx:          # a back edge: once more through the loop body, same state
            seen = len(bad)
            for _ in range(ahead + 1):                               # (a stream that runs `ahead` groups ahead shows a miscount one pass later)
                for k in range(labels[mb.group(1)], idx + 1):
                    step(k, body[k])
            for k in range(seen, len(bad)):
                bad[k] = (bad[k][0], bad[k][1] + ' (second pass through the loop: the back edge)', bad[k][2])
    return bad


def scan_text(text):
    """[(kernel, store, overwriting instruction)] of one assembly file, and the number of wide buffer stores seen."""
    bad, n = [], 0
    for name, k in kernels_of(text).items():
        n += sum(1 for ins in k['body'] if re.match(r'buffer_store_dwordx[34]', ins))
        bad += [(name, a, b) for a, b in store_hazards(k['body'])]
        if '_h3' in name and any(ins.startswith('buffer_load') and ins.rstrip().endswith(' lds') for ins in k['body']):
            # the wide kernel: four buffers, the group requested in tap T is waited for at the end of tap T + 1 (one group in flight across a
            # barrier); its prologue requests three taps back to back and drains them
            # (the pipe kernel, kernels_conv_f16x3_pipe.hip: the same protocol with 8 KiB blocks -- two DMAs per helper wave and pass-tap)
            wide, pipe = '_h3w' in name, '_h3p' in name
            bad += [(name, 'LDS-DMA order: ' + why, ins)
                    for _, why, ins in dma_order_violations(k['body'], group=2 if pipe else 4, ahead=1 if wide or pipe else 0,
                                                            primed=1 if wide or pipe else 0, labels=k['labels'])]
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

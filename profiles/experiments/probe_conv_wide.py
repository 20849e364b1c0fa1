#!/usr/bin/env python3
"""Round 6: the WIDE f16x3 conv3x3 kernel (kernels_conv_f16x3_wide.hip) against the narrow one (kernels_conv_f16x3.hip), same process image,
the variant chosen by the developer knob PNP_CONV_WIDE (read once per process: this script re-runs itself per variant).
  check:  every combination of activation formats x {skip, no skip} x {relu, none} x {bias, none} on shapes with ragged edges and C in
          {64, 128, 256}: sha256 of the result per case; the parent compares the two variants (bit-equality is the bar)
  time:   [n, 64, 128, 128] split -> split (FFDNet body), [64, 64, 256, 256], [64, 128, 128, 128], [64, 256, 64, 64], [64, 512, 32, 32]
usage (GPU box): python3 profiles/experiments/probe_conv_wide.py [check|time|both] [n=64]"""
import ctypes as C, hashlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def child(what, n):
    import torch
    from pnp_admm_cnc_mri_amd import _lib, denoisers as D
    L = _lib.lib()
    dev = torch.device('cuda', 0)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    out = {'check': {}, 'time': {}}

    def pack(Cc, seed):
        g = torch.Generator(device='cuda').manual_seed(seed)
        w = torch.randn(Cc, Cc, 3, 3, device=dev, generator=g) * (2.0 / (9 * Cc)) ** 0.5
        wp = torch.empty(9 * Cc * Cc, device=dev)
        _lib.check(L.pnp_conv3x3_pack_f16x3(s, p(w), p(wp), Cc))
        return wp
    if what in ('check', 'both'):
        for (nn, Cc, H, W) in ((3, 64, 37, 53), (2, 64, 16, 16), (1, 64, 5, 130), (2, 128, 40, 24), (1, 256, 17, 33), (5, 64, 64, 64)):
            wp = pack(Cc, 1)
            g = torch.Generator(device='cuda').manual_seed(2)
            x = torch.randn(nn, H, W, Cc, device=dev, generator=g)
            k = torch.randn(nn, H, W, Cc, device=dev, generator=g)
            b = torch.randn(Cc, device=dev, generator=g) * 0.1
            xs, ks = D.split_activations(x), D.split_activations(k)
            for fmt in range(8):
                for skip in (0, 1):
                    if (fmt & 2) and not skip:
                        continue
                    for relu in (0, 1):
                        for bias in (0, 1):
                            y = torch.full_like(x, 7.0)
                            _lib.check(L.pnp_conv3x3_nhwc_f16x3_fmt(s, p(xs if fmt & 1 else x), p(wp), p(b) if bias else None,
                                                                    p((ks if fmt & 2 else k)) if skip else None, p(y), nn, Cc, H, W, relu, 1, fmt))
                            torch.cuda.synchronize()
                            out['check']['%s fmt%d skip%d relu%d bias%d' % ((nn, Cc, H, W), fmt, skip, relu, bias)] = \
                                hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()[:16]
    if what in ('time', 'both'):
        for (nn, Cc, H, W, fmt, skip) in ((n, 64, 128, 128, 5, 0), (n, 64, 128, 128, 0, 0), (n, 64, 256, 256, 5, 0), (n, 128, 128, 128, 7, 1),
                                          (n, 256, 64, 64, 7, 1), (n, 512, 32, 32, 7, 1), (1, 64, 256, 256, 5, 0), (1, 64, 128, 128, 5, 0)):
            wp = pack(Cc, 1)
            x = torch.relu(torch.randn(nn, H, W, Cc, device=dev))
            k = torch.randn(nn, H, W, Cc, device=dev)
            b = torch.randn(Cc, device=dev) * 0.1
            xin = D.split_activations(x) if fmt & 1 else x
            kin = (D.split_activations(k) if fmt & 2 else k) if skip else None
            y = torch.empty_like(x)
            run = lambda: _lib.check(L.pnp_conv3x3_nhwc_f16x3_fmt(s, p(xin), p(wp), p(b), p(kin), p(y), nn, Cc, H, W, 1, 1, fmt))
            for _ in range(5):
                run()
            torch.cuda.synchronize()
            reps = 40 if nn > 1 else 200
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            flop = 2.0 * nn * H * W * Cc * Cc * 9
            out['time']['[%d, %d, %d, %d] fmt %d skip %d' % (nn, Cc, H, W, fmt, skip)] = [round(ms, 4), round(3 * flop / ms / 1e9 / 2.5e6, 3)]
    print('RESULT ' + json.dumps(out), flush=True)


if __name__ == '__main__':
    if os.environ.get('PROBE_CHILD'):
        child(sys.argv[1], int(sys.argv[2]))
        sys.exit(0)
    what = sys.argv[1] if len(sys.argv) > 1 else 'both'
    n = sys.argv[2] if len(sys.argv) > 2 else '64'
    res = {}
    for mode in ('0', '1', '0', '1') if what != 'check' else ('0', '1'):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), what, n], env=dict(os.environ, PROBE_CHILD='1', PNP_CONV_WIDE=mode),
                           capture_output=True, text=True, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith('RESULT ')]
        if r.returncode or not line:
            print('variant %s FAILED rc %d\n%s\n%s' % (mode, r.returncode, r.stdout[-2000:], r.stderr[-3000:]))
            sys.exit(1)
        res.setdefault(mode, []).append(json.loads(line[0][7:]))
    a, b = res['0'][0]['check'], res['1'][0]['check']
    bad = [k for k in a if a[k] != b.get(k)]
    print('check: %d cases, %d differ between narrow (PNP_CONV_WIDE=0) and wide (1)' % (len(a), len(bad)))
    for k in bad[:40]:
        print('   DIFF', k)
    for k in res['0'][0]['time']:
        print('%-38s narrow %s   wide %s   (ms; two runs each)' % (
            k, ' / '.join('%.4f' % r['time'][k][0] for r in res['0']), ' / '.join('%.4f' % r['time'][k][0] for r in res['1'])))
    sys.exit(1 if bad else 0)

# Layout search over pitch / mirror offset / XOR swizzle of the slice kernel's transposition buffer with the model of lds_bank_model.py.
import sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
from lds_bank_model import cycles
def make_addr(P, M, mult, rmod, mask, mshift):
    def A(r, c, mirror):
        sig = (mult * (r % rmod)) & mask
        if mirror: return r * P + M + (((c - mshift) & 63) ^ sig)
        return r * P + (c ^ sig)
    return A
def evaluate(A, swap_regs):
    tot = {}
    def add(name, kind, addrs): tot[name] = tot.get(name, 0) + cycles(kind, addrs)
    for wv in range(8):
        for h in range(2):
            for j in range(16):
                addrs = [8 * A(((l & 15) >> 1) + 8 * j, 32 * h + 4 * wv + (l >> 4), l & 1) for l in range(64)]
                add('T1 read', 'read_b64', addrs); add('T2 store', 'write_b64', addrs)
    for wv in range(8):
        for s in range(4):
            for m in range(4):
                addrs = [8 * A(32 * s + 4 * wv + (l >> 4), (l & 15) + 16 * m, 0) for l in range(64)]
                add('T1 store direct', 'write_b64', addrs); add('T2 load direct', 'read_b64', addrs)
            for jj in range(4):
                addrs = []
                for l in range(64):
                    t = l & 15
                    c = (16 * (jj + 1) - t) if jj < 3 else ((64 - t) if t else 0)
                    addrs.append(8 * A(32 * s + 4 * wv + (l >> 4), c, 1))
                if swap_regs:
                    add('T1 store mirror', 'write_b64', addrs); add('T2 load mirror', 'read_b64', addrs)
                else:
                    for part in (0, 4):
                        add('T1 store mirror', 'write_b32', [a + part for a in addrs]); add('T2 load mirror', 'read_b32', [a + part for a in addrs])
    return tot
if __name__ == '__main__':
    res = []
    for P in range(128, 161):
        for M in range(64, P - 63):
            for (mult, rmod, mask) in [(0,1,0),(1,8,15),(2,8,15),(1,16,15),(3,8,15),(5,8,15)]:
                for ms in (0, 1):
                    A = make_addr(P, M, mult, rmod, mask, ms)
                    ok = True
                    for r in range(16):
                        seen = set()
                        for c in range(64):
                            for mir in (0, 1):
                                a = A(r, c, mir) - r * P
                                if a < 0 or a >= P or a in seen: ok = False
                                seen.add(a)
                    if not ok: continue
                    t = evaluate(A, True)
                    res.append((sum(t.values()), P, M, mult, rmod, mask, ms, t))
    res.sort(key=lambda x: (x[0], x[1]))
    for r in res[:8]: print(r)

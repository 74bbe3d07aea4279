"""LDS-array cycles of a wave64 access on gfx950, by the lane-group / bank rules of
/opt/skills/guides/MI355X_MICROARCH.md (LDS section): used to choose the layouts of the fused level
kernel (gslam_amd/csrc/gklt_pyramid.hip).  addr: byte address per lane (None = inactive)."""
G128 = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
        list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
G128 = G128 + [[l + 32 for l in g] for g in G128]
GROUPS = {
    "read_b32": ([list(range(0, 32)), list(range(32, 64))], 32, 1),
    "read_b64": ([list(range(0, 32)), list(range(32, 64))], 64, 2),
    "read_b128": (G128, 64, 4),
    "write_b32": ([list(range(0, 32)), list(range(32, 64))], 32, 1),
    "write_b64": ([list(range(16 * g, 16 * g + 16)) for g in range(4)], 32, 2),
    "write_b128": ([list(range(8 * g, 8 * g + 8)) for g in range(8)], 32, 4),
}


def cycles(kind, addr):
    groups, mod, nd = GROUPS[kind]
    tot = 0
    for g in groups:
        banks = {}
        for l in g:
            if addr[l] is None:
                continue
            for d in range(nd):
                a = addr[l] // 4 + d
                banks.setdefault(a % mod, set()).add(a)
        tot += max([len(v) for v in banks.values()], default=0)
    return tot


def ideal(kind):
    return len(GROUPS[kind][0])


if __name__ == "__main__":
    import sys
    LNT = 256
    R0, RP, RG, LTH, TW = 2, 5, 3, 32, 64
    HI = max(RP, RG); HIP = (HI + 3) // 4 * 4; AW = TW + 2 * HIP; IHh = LTH + 2 * HI
    R0P = (R0 + 3) // 4 * 4; SW = AW + 2 * R0P; SH = IHh + 2 * R0
    if len(sys.argv) > 1: SW = int(sys.argv[1])
    if len(sys.argv) > 2: AW = int(sys.argv[2])
    print("AW", AW, "SW", SW, "SH", SH, "IHh", IHh)

    def run(name, kind, fn, ntask):
        c = i = 0
        for base in range(0, ntask, 64):
            addr = [fn(base + l) if base + l < ntask else None for l in range(64)]
            c += cycles(kind, addr); i += sum(1 for g in GROUPS[kind][0] if any(addr[l] is not None for l in g))
        print("%-34s %-10s cycles %5d ideal %5d" % (name, kind, c, i))
        return c, i
    T = [0, 0]
    def acc(r):
        T[0] += r[0]; T[1] += r[1]
    GPR = AW // 4
    for q in range(3):
        acc(run("phase2 read S q=%d" % q, "read_b128", lambda t: ((t // GPR) * SW + 4 * (t % GPR) + 4 * q) * 4, SH * GPR))
    acc(run("phase2 write A", "write_b128", lambda t: ((t // GPR) * AW + 4 * (t % GPR)) * 4, SH * GPR))
    H = AW // 2; STR = LNT // H; SR = (IHh + STR - 1) // STR
    for i in range(SR + 2 * R0):
        acc(run("phase3 read A i=%d" % i, "read_b64", lambda t: (min((t // H) * SR + i, SH - 1) * H + t % H) * 8, STR * H))
    for j in range(SR):
        acc(run("phase3 write I j=%d" % j, "write_b64", lambda t: (((t // H) * SR + j) * H + t % H) * 8 if (t // H) * SR + j < IHh else None, STR * H))
    for q in range(5):
        acc(run("phase4 read I q=%d" % q, "read_b128", lambda t: ((t >> 4) * AW + 4 * (t & 15) + 4 * q) * 4, IHh * 16))
    acc(run("phase4 write rowp", "write_b128", lambda t: ((t >> 4) * 64 + 4 * (t & 15)) * 4, IHh * 16))
    print("total", T)

    # ---- with the wave's lanes permuted so that each 16-lane group of ds_read_b128 takes 16
    # consecutive tasks
    slot = [0] * 64
    for h in range(2):
        for gi, g in enumerate(G128[2 * h:2 * h + 2]):
            for k, l in enumerate(g):
                slot[l] = 32 * h + 16 * gi + k
    print("slot", slot)
    STORE = {"write_b32": 4, "write_b64": 6, "write_b128": 13}
    def run2(name, kind, fn, ntask):
        c = 0
        for base in range(0, ntask, 64):
            addr = [fn(base + slot[l]) if base + slot[l] < ntask else None for l in range(64)]
            cc = cycles(kind, addr)
            if kind in STORE: cc = max(cc, STORE[kind])
            c += cc
        print("%-34s %-10s cycles %5d" % (name, kind, c))
    for q in range(3):
        run2("perm phase2 read S q=%d" % q, "read_b128", lambda t: ((t // GPR) * SW + 4 * (t % GPR) + 4 * q) * 4, SH * GPR)
    run2("perm phase2 write A", "write_b128", lambda t: ((t // GPR) * AW + 4 * (t % GPR)) * 4, SH * GPR)
    for q in range(5):
        run2("perm phase4 read I q=%d" % q, "read_b128", lambda t: ((t >> 4) * AW + 4 * (t & 15) + 4 * q) * 4, IHh * 16)
    run2("perm phase4 write rowp", "write_b128", lambda t: ((t >> 4) * 64 + 4 * (t & 15)) * 4, IHh * 16)
    slot = list(range(64))
    run2("noperm phase4 write rowp", "write_b128", lambda t: ((t >> 4) * 64 + 4 * (t & 15)) * 4, IHh * 16)
    run2("noperm phase2 write A", "write_b128", lambda t: ((t // GPR) * AW + 4 * (t % GPR)) * 4, SH * GPR)

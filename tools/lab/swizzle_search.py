"""Search for the LDS chunk swizzles of the 32-row W/H-phase kernel (kernels_bf16.hip, xyt32).

Y image: [64 factor rows][128 B of bf16] (hi and lo images alike), 16-byte chunk c of row r at position
c ^ f(r), f XOR-linear in the row bits.  It must be conflict free for
  R1  ds_read_b128 row reads of the 32x32x16 B operand (A-product),
  R2  ds_read_b64_tr_b16 transposed reads of the 32x32x16 A operand (residual product),
  R3  ds_read_b128 row reads of the 16x16x32 operands (Gram by-product).
Bank rules and lane groups: MI355X_MICROARCH.md, LDS table.  XOR with a chunk index that is constant over
a lane group is a bijection on positions, so one representative (column half, k-step, tile) per read kind
is enough."""
import itertools

B128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
               [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
B128_GROUPS += [[l + 32 for l in g] for g in B128_GROUPS]
B64_GROUPS = [list(range(32)), list(range(32, 64))]


def conflicts(addr_of_lane, groups, width):
    extra = 0
    for g in groups:
        cnt = {}
        for l in g:
            a = addr_of_lane(l)
            for d in range(width // 4):
                cnt.setdefault((a // 4 + d) % 64, set()).add(a // 4 + d)
        extra += max(len(v) for v in cnt.values()) - 1
    return extra


def fmap(masks, r):
    return sum(((bin(r & m).count("1") & 1) << i) for i, m in enumerate(masks))


def y_r1(f):
    return conflicts(lambda l: 128 * (l & 31) + 16 * ((l >> 5) ^ f[l & 31]), B128_GROUPS, 16)


def y_r2(f):
    def addr(l):
        G, q, p = l >> 4, (l >> 2) & 3, l & 3
        row = 8 * (G >> 1) + q
        return 128 * row + 16 * ((2 * (G & 1) + (p >> 1)) ^ f[row]) + 8 * (p & 1)
    return conflicts(addr, B64_GROUPS, 8)


def y_r3(f):
    return conflicts(lambda l: 128 * (l & 15) + 16 * ((l >> 4) ^ f[l & 15]), B128_GROUPS, 16)


def y_old_tr(f):   # the 16x16x32 transposed reads of the existing kernel (tro[]): rows 8 g + q, columns 16 e + 4 p
    def addr(l):
        x, g = l & 15, l >> 4
        q, p = x >> 2, x & 3
        row = 8 * g + q
        return 128 * row + 16 * ((p >> 1) ^ f[row]) + 8 * (p & 1)
    return conflicts(addr, B64_GROUPS, 8)


def v_score(f):
    """V tile [32 rows][256 B of f32]: 16 chunks per row, position c ^ f(row)."""
    tot = 0
    for e in range(2):      # A layout: chunk 2 b + e (+ const)
        tot += conflicts(lambda l: 256 * (l & 31) + 16 * ((2 * (l >> 5) + e) ^ f[l & 31]), B128_GROUPS, 16)
    tot += conflicts(lambda l: 256 * (l & 31) + 16 * ((l >> 5) ^ f[l & 31]), B128_GROUPS, 16)      # D layout
    return tot


if __name__ == "__main__":
    good = []
    for masks in itertools.product(range(32), repeat=3):
        f = [fmap(masks, r) for r in range(32)]
        if y_r1(f):
            continue
        if y_r2(f):
            continue
        good.append((y_r3(f) + y_old_tr(f), masks))
    good.sort()
    print("Y: conflict-free for R1 and R2:", len(good), "best by R3 + old transposed reads:", good[:6])
    old = [(((r >> 1) & 1) << 1) | (((r >> 3) & 1) << 2) for r in range(32)]
    print("old yswz: R1", y_r1(old), "R2", y_r2(old), "R3", y_r3(old), "old tr", y_old_tr(old))
    print("V with r & 15:", v_score([r & 15 for r in range(32)]))

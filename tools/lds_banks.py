"""LDS bank-conflict model for gfx950 (rules from /opt/skills/guides/MI355X_MICROARCH.md, section LDS):
lane groups per instruction, bank = (byte_addr/4) mod 64 for ds_read_b64 / b128 / b64_tr_b16.
cost(group) = max over banks of the number of distinct dwords mapped to it; 1 = conflict-free.
Used to design the LDS images of the two GEMM kernels."""

G128 = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
        list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
        list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
        list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]
G64 = [list(range(0, 32)), list(range(32, 64))]


def cost(addr_of_lane, width, groups, nbanks=64):
    worst = []
    for g in groups:
        per_bank = {}
        for l in g:
            a = addr_of_lane(l)
            assert a % min(width, 16) == 0 or width == 8 and a % 8 == 0, (l, a)
            for dw in range(a // 4, (a + width) // 4):
                per_bank.setdefault(dw % nbanks, set()).add(dw)
        worst.append(max(len(s) for s in per_bank.values()))
    return worst


def b128(addr_of_lane):
    return cost(addr_of_lane, 16, G128)


def tr_b64(addr_of_lane):
    return cost(addr_of_lane, 8, G64)


if __name__ == "__main__":
    # forward GEMM operand image: [rows][64 halves] = 128-B rows, chunk' = chunk ^ (row & 7)
    for kk in range(2):
        print("fwd A frag kk=%d" % kk,
              b128(lambda l: (l & 15) * 128 + ((((kk * 4 + (l >> 4)) ^ ((l & 15) & 7))) << 4)))

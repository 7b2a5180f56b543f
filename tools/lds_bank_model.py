#!/usr/bin/env python3
"""Host model of LDS bank conflicts for the paged-decode kernel's tile image
(minivllm_amd/csrc/attn_decode.hip, head_dim 128: 32 rows of 16 chunks of 16 bytes per wave).

The LDS services a wave64 access in fixed lane groups, one cycle per group when every bank is
asked for one address at most; each extra address on a busy bank costs one more cycle
(SQ_LDS_BANK_CONFLICT counts exactly those).  Groups and bank widths:

  ds_read_b128        4 groups of 16 lanes {0-3,12-15,20-27} {4-11,16-19,28-31} (+32), 64 banks
  ds_read_b64_tr_b16  2 groups of 32 lanes, 64 banks
  ds_write_b128       8 groups of 8 consecutive lanes, 32 banks

For the XOR pattern the kernel started with, the model gives 64 extra cycles per 32-token tile
(32 from the K fragment reads, 32 from the transposed V reads), which is what rocprofv3 counted:
4,194,304 conflict cycles per dispatch = 64 x 65,536 wave-tiles at 256 sequences x 1024 tokens.
For the pattern in the kernel now it gives 0, and so does the counter
(profiles/r1_fp8_decode_kernel_pmc_counters.txt).

    python tools/lds_bank_model.py
"""
CPR = 16                                  # 16-byte chunks per 256-byte row


def x_first(row):                         # the pattern shared with the prefill kernel (common.h)
    return (((row & 3) << 2) | ((row >> 2) & 3)) & (CPR - 1)


def x_decode(row):                        # dec_tile_off in attn_decode.hip
    return (((row & 7) << 1) | ((row >> 2) & 1)) & (CPR - 1)


def off(x, row, ch):
    return row * CPR * 16 + ((ch ^ x(row)) << 4)


B128_READ_GROUPS = [
    list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
    list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
    list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
    list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64)),
]
HALVES = [list(range(32)), list(range(32, 64))]
OCTETS = [list(range(8 * g, 8 * g + 8)) for g in range(8)]


def extra_cycles(addr_of_lane, groups, width, nbanks):
    extra = 0
    for g in groups:
        banks = {}
        for lane in g:
            a = addr_of_lane[lane]
            for w in range(width // 4):
                banks.setdefault((a // 4 + w) % nbanks, set()).add(a)
        extra += max(len(v) for v in banks.values()) - 1
    return extra


def per_tile(x):
    """Extra LDS cycles per 32-token tile and wave, by access kind."""
    k = sum(extra_cycles([off(x, 16 * sub + (l & 15), 4 * s + (l >> 4)) for l in range(64)],
                         B128_READ_GROUPS, 16, 64)
            for sub in range(2) for s in range(4))
    v = 0
    for n in range(8):
        for hi in (0, 16):
            addrs = []
            for l in range(64):
                c16, q4 = l & 15, l >> 4
                addrs.append(off(x, hi + 4 * q4 + (c16 >> 2), 2 * n + ((c16 & 3) >> 1)) + (c16 & 1) * 8)
            v += extra_cycles(addrs, HALVES, 8, 64)
    w16 = sum(extra_cycles([off(x, (i * 64 + l) // 16, (i * 64 + l) % 16) for l in range(64)],
                           OCTETS, 16, 32) for i in range(8))
    w8 = 0
    for i in range(4):                    # fp8 cache: lane -> (row, chunk) as in the kernel
        for half in (0, 1):
            addrs = []
            for l in range(64):
                ch = ((l >> 3) & 1) * 4 + (l & 3)
                row = i * 8 + ((l >> 2) & 1) * 4 + ((l >> 4) & 3)
                addrs.append(off(x, row, 2 * ch + half))
            w8 += extra_cycles(addrs, OCTETS, 16, 32)
    return {"k_reads": k, "v_reads": v, "stores_16bit": w16, "stores_fp8": w8}


if __name__ == "__main__":
    for name, x in (("first pattern", x_first), ("decode pattern", x_decode)):
        print(f"{name:15s}", per_tile(x))

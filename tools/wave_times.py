#!/usr/bin/env python3
"""When and where every wave of ssa_k_verify ran (diagnostic build: tools/build_variants.sh wt:"-DSSA_WAVE_TIMES").
    SSA_LIB=build/variants/wt.so python3 tools/wave_times.py [n]
Prints the kernel's span, the distribution of wave start times and durations, and the same per XCD."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
import schnorr_sig_amd as ssa  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    dev = torch.device("cuda", 0)
    eng = ssa.Engine(0)
    sigs, pks, msgs, g = bench.gen_batch(torch, eng, dev, n, 0x5C4E0222)
    status = torch.empty(n, dtype=torch.uint8, device=dev)
    nfail = torch.zeros(1, dtype=torch.int64, device=dev)
    for _ in range(2):
        eng.verify_many_device(sigs.data_ptr(), pks.data_ptr(), msgs.data_ptr(), n, 80, status.data_ptr(), nfail.data_ptr(),
                               check_torsion=False, sig_flag_byte=True)
        eng.sync()
    lib = C.CDLL(os.environ["SSA_LIB"])
    if "--phases" in sys.argv:
        # where an ordinary wave's time goes, against where its instructions are (tail_plan's counts: table 57 k,
        # 50 windows of 16.9 k, comb + comparison 45 k)
        ph = np.zeros((1 << 16, 4), dtype=np.uint64)
        assert lib.ssa_debug_phase_times(ph.ctypes.data_as(C.c_void_p), C.c_size_t(1 << 16)) == 0
        ph = ph[(ph[:, 0] != 0) & (ph[:, 3] != 0) & (ph[:, 3] + np.uint64(20_000_000) >= ph[:, 3].max())].astype(np.float64) / 100.0
        d = np.diff(ph, axis=1)
        tot = d.sum(axis=1)
        share = (57.0, 847.4, 45.0)
        print("%d ordinary waves, mean life %.1f us" % (len(ph), tot.mean()))
        for k, nm in enumerate(("start -> table built", "ladder (50 windows)", "comb + comparison")):
            print("  %-22s mean %7.1f us = %5.2f %% of the wave's life; %5.2f %% of its instructions; us per 1000 instructions %.3f"
                  % (nm, d[:, k].mean(), 100 * d[:, k].mean() / tot.mean(), 100 * share[k] / sum(share), d[:, k].mean() / share[k]))
        return
    cap = 1 << 16
    raw = np.zeros((cap, 3), dtype=np.uint64)
    assert lib.ssa_debug_wave_times(raw.ctypes.data_as(C.c_void_p), C.c_size_t(cap)) == 0
    # the waves of the LAST launch (the end game adds piece workgroups to the grid: more waves than n / 64)
    keep = (raw[:, 1] != 0) & (raw[:, 0] + np.uint64(20_000_000) >= raw[:, 1].max())
    buf = raw[keep]
    nw = len(buf)
    if len(sys.argv) > 2:
        np.save(sys.argv[2], buf)
    t0 = buf[:, 0].min()
    st = (buf[:, 0] - t0).astype(np.float64) / 100.0      # microseconds (100 MHz)
    en = (buf[:, 1] - t0).astype(np.float64) / 100.0
    dur = en - st
    xcc = (buf[:, 2] >> np.uint64(32)).astype(np.int64) & 0xf
    hw = buf[:, 2].astype(np.int64) & 0xffffffff
    print("n = %d, %d waves; kernel span %.1f us; wave duration mean %.1f us, min %.1f, p50 %.1f, p99 %.1f, max %.1f"
          % (n, nw, en.max(), dur.mean(), dur.min(), np.median(dur), np.percentile(dur, 99), dur.max()))
    order = np.argsort(st)
    # generations: waves sorted by start time, in groups of the resident capacity (8192 waves = 2 per SIMD)
    cap = 256 * 4 * 2
    for gidx in range(0, nw, cap):
        sel = order[gidx:gidx + cap]
        print("  generation %d: starts %.1f .. %.1f us, ends %.1f .. %.1f us, duration mean %.1f (min %.1f max %.1f)"
              % (gidx // cap, st[sel].min(), st[sel].max(), en[sel].min(), en[sel].max(), dur[sel].mean(), dur[sel].min(), dur[sel].max()))
    for x in sorted(set(xcc.tolist())):
        sel = xcc == x
        print("  XCC %d: %5d waves, duration mean %.1f us, last end %.1f us" % (x, sel.sum(), dur[sel].mean(), en[sel].max()))
    # busy-slot integral: how much of (span x capacity) is covered by waves
    print("  sum of wave durations / (span x %d slots) = %.4f" % (cap, dur.sum() / (en.max() * cap)))
    short = dur < 0.6 * np.median(dur)
    if short.any():
        print("  %d short waves (pieces of the end game): duration mean %.1f us; %d of them end in the last %.0f us"
              % (short.sum(), dur[short].mean(), int((short & (en > en.max() - 1500)).sum()), 1500))
    last = order[-cap:]
    print("  the last %d waves to start: starts %.1f .. %.1f us, ends %.1f .. %.1f us" % (len(last), st[last].min(), st[last].max(),
                                                                                        en[last].min(), en[last].max()))
    cu = (hw >> 8) & 0xf
    se = (hw >> 13) & 0x7
    print("  hw id fields seen: cu %s se %s" % (sorted(set(cu.tolist())), sorted(set(se.tolist()))))


if __name__ == "__main__":
    main()

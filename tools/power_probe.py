import sys, os, time, subprocess, threading
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
import schnorr_sig_amd as ssa
dev = torch.device("cuda", 0)
eng = ssa.Engine(0)
n = 1 << 20
sigs, pks, msgs, g = bench.gen_batch(torch, eng, dev, n, 0x5C4E0222)
status = torch.empty(n, dtype=torch.uint8, device=dev)
nfail = torch.zeros(1, dtype=torch.int64, device=dev)
felts = torch.randint(0, 2**62, (n, 25), dtype=torch.int64, device=dev)
out = torch.empty((n, 4), dtype=torch.int64, device=dev)
def smi(tag):
    r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp"], capture_output=True, text=True).stdout
    keep = [l.strip() for l in r.splitlines() if ("Power" in l or "sclk" in l or "junction" in l.lower())]
    print(tag, " | ".join(keep)[:600], flush=True)
smi("idle")
for name, fn in (("hash", lambda: eng.hash_message_many_device(sigs.data_ptr(), pks.data_ptr(), msgs.data_ptr(), n, 80, out.data_ptr()) if hasattr(eng, "hash_message_many_device") else None),
                 ("verify_step", lambda: eng.verify_many_device(sigs.data_ptr(), pks.data_ptr(), msgs.data_ptr(), n, 80, status.data_ptr(), nfail.data_ptr(), check_torsion=False, sig_flag_byte=True))):
    if fn() is None and name == "hash":
        pass
    t_end = time.time() + 6
    k = 0
    while time.time() < t_end:
        for _ in range(20):
            fn()
        k += 1
        if k % 3 == 0:
            smi(name)
    eng.sync()

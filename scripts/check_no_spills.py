"""Build check: k_dense_fold loads through inline asm (global_load_dwordx4 + hand-counted s_waitcnt vmcnt),
which is only correct while the buffers' registers are never spilled or copied between issue and wait.  Fails the build when
any instantiation of the named kernels spills VGPRs or uses scratch (SGPR spills go to lanes of a VGPR of their own
and do not move the buffers).
    python3 scripts/check_no_spills.py <-Rpass-analysis=kernel-resource-usage log> <kernel name substring> ..."""
import re
import sys

log, names = sys.argv[1], sys.argv[2:]
cur, seen, bad = None, 0, []
for line in open(log, errors="replace"):
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        seen += any(n in cur for n in names)
        continue
    m = re.search(r"(VGPRs Spill|ScratchSize \[bytes/lane\]): (\d+)", line)
    if m and cur and any(n in cur for n in names) and int(m.group(2)) != 0:
        bad.append(f"{cur}: {m.group(1)} = {m.group(2)}")
if not seen:
    sys.exit(f"check_no_spills: no kernel matching {names} in {log}")
if bad:
    sys.exit("check_no_spills: register spills in a kernel whose inline-asm loads forbid them:\n  " + "\n  ".join(bad))
print(f"check_no_spills: {seen} instantiation(s) of {names}: no spills, no scratch")

"""Static check of the inline-asm MFMA statements of csrc/prefill_attn64.hip (r04).

The compiler does not know that those asm statements are MFMAs, so it inserts none of the wait states the matrix pipe needs
around VGPR operands it copies on its own.  This script compiles the file to ISA and, in file order (a linear approximation of
the control flow), flags
  (a) a VALU write to a VGPR less than 2 wait states before an MFMA that reads it as SrcC, and
  (b) a non-MFMA read of an MFMA's VGPR result less than 18 wait states / 4 MFMAs after it.
Exit code 1 when anything is flagged.  Usage: python tools/check_mfma_hazards.py [file.s]"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "glm4_moe_nano-vllm_amd", "csrc", "prefill_attn64.hip")


def regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"-?\|?v(\d+)\|?", tok)
    return {int(m.group(1))} if m else set()


def main():
    if len(sys.argv) > 1:
        text = open(sys.argv[1]).read()
    else:
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "k.s")
            subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-ffp-contract=off", "--offload-arch=gfx950",
                            "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S", "-o", out, SRC],
                           check=True, capture_output=True)
            text = open(out).read()
    ins = []
    for ln in text.splitlines():
        t = ln.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        t = t.split(";")[0].strip()
        parts = t.replace(",", " ").split()
        ins.append((parts[0], parts[1:]))
    bad = 0
    for i, (op, a) in enumerate(ins):
        if not op.startswith("v_mfma"):
            continue
        dst, srcc = regs(a[0]), regs(a[3]) if len(a) > 3 else set()
        # (a) VALU write right in front of the SrcC read
        ws = 0
        for j in range(i - 1, max(i - 4, -1), -1):
            o, b = ins[j]
            if o == "s_nop":
                ws += int(b[0]) + 1
                continue
            if ws >= 2:
                break
            if o.startswith("v_") and not o.startswith("v_mfma") and b and regs(b[0]) & srcc:
                print(f"(a) VALU write {o} {' '.join(b)} {ws} wait states before MFMA #{i} reading it as SrcC: {' '.join(a)}")
                bad += 1
            ws += 1
        # (b) early read of the VGPR result
        if not dst:
            continue
        ws, mf = 0, 0
        for j in range(i + 1, min(i + 40, len(ins))):
            o, b = ins[j]
            if ws >= 18 or mf >= 4:
                break
            if o == "s_nop":
                ws += int(b[0]) + 1
                continue
            if o.startswith("v_mfma"):
                mf += 1
                ws += 4
                continue
            srcs = set()
            for tok in (b[1:] if o.startswith(("v_", "ds_", "global_", "buffer_")) and not o.startswith(("global_store", "ds_write")) else b):
                srcs |= regs(tok)
            if srcs & dst:
                print(f"(b) {o} {' '.join(b)} reads the result of MFMA #{i} ({' '.join(a)}) after {ws} wait states, {mf} MFMAs")
                bad += 1
            ws += 1
    print(f"{sum(1 for o, _ in ins if o.startswith('v_mfma'))} MFMAs checked, {bad} findings")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

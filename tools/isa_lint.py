"""Hazard lint of the gfx950 device assembly the library is built from (mocopci_amd/csrc/isa/*.s, `make isa`).

Why: the compiler's hazard recogniser and wait-count pass do not look INTO inline asm.  An inline-asm VALU instruction is
not a "VALU" to GCNHazardRecognizer: it gets none of the wait states an MFMA -> VALU or transcendental -> VALU read needs,
and as a producer it is not counted for the VALU -> v_readlane / v_permlane*_swap rules; an inline-asm LDS atomic is
invisible to SIInsertWaitcnts.  The kernels keep a few raw instructions (v_min/v_max without the canonicalising v_max x,x the
compiler adds for values it cannot prove quiet, v_max_f32_dpp with bank masks, v_min/v_max_f64 on sort keys), so this tool
checks the FINAL instruction stream instead of trusting their surroundings: for every instruction of the opcodes the asm
statements emit -- wherever it came from -- it walks backwards over every control-flow path and requires the wait states of
the gfx940/gfx950 hazard table between it and the producers of its sources:

    producer                          consumer                                   wait states
    v_mfma_* (N passes)               any checked VALU reading its result        N + 3 (+1 on gfx950 when N > 2)
    v_exp/log/rcp/rsq/sqrt/sin/cos    any checked VALU                           1
    any VALU                          *_dpp reading its result                   2
    checked opcode (maybe from asm)   v_readlane / v_readfirstlane               1
    checked opcode (maybe from asm)   v_permlane16/32_swap                       2
    v_pk_*_f32                        anything                                   NOT ALLOWED in the library (DESIGN.md section 6:
                                                                                 the compiler covers VALU consumers with one wait
                                                                                 state only by an accident of modifier bits, DS /
                                                                                 VMEM consumers with none; hardware needs more)
    ds_max_u64 / ds_max_rtn_u64       next s_barrier                             an s_waitcnt lgkmcnt(..) in between

usage: python tools/isa_lint.py <file.s ...>      exit code 1 and one line per finding when something is uncovered."""
import re
import sys

CHECKED = ("v_max_f32", "v_min_f32", "v_max3_f32", "v_min3_f32", "v_max_f64", "v_min_f64")
TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")
BRANCH = ("s_branch", "s_cbranch", "s_setpc", "s_endpgm", "s_swappc")
MAX_LOOKBACK = 24


def mfma_wait(op):
    """Wait states a matrix-core result needs before a VALU may read it (LLVM GCNHazardRecognizer::checkMAIVALUHazards, gfx950):
    f32-input MFMA ("SGEMM" class, e.g. v_mfma_f32_32x32x2_f32): passes + 2; XDL (bf16 / f16 / i8 / fp8 input): passes + 3, + 1
    beyond two passes.  One pass = 4 cycles; passes follow from the tile's flops over the pipe's rate."""
    m = re.search(r"_(\d+)x(\d+)x(\d+)_?(\w*)$", op)
    if not m:
        return 20
    mm, nn, kk, typ = int(m.group(1)), int(m.group(2)), int(m.group(3)), m.group(4)
    flops = 2 * mm * nn * kk
    if typ.startswith("f32") or typ in ("xf32",):
        passes = max(flops // 64 // 4, 2)       # 64 flop/clk/SIMD on the f32-input path
        return passes + 2
    rate = 1024 if kk >= 16 and mm * nn >= 256 else 512   # gfx950 double-K tiles run at twice the legacy rate
    passes = max(flops // rate // 4, 2)
    return passes + 3 + (1 if passes > 2 else 0)


def regs(tok):
    tok = tok.strip()
    m = re.match(r"^-?\|?v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"^-?\|?v(\d+)\b", tok)
    if m:
        return {int(m.group(1))}
    return set()


class Ins:
    __slots__ = ("op", "dst", "src", "line", "text", "asm")


def parse(path):
    """-> (instructions, label -> index of the first instruction after it)"""
    ins, labels, pending = [], {}, []
    in_asm = False
    for ln, raw in enumerate(open(path), 1):
        text = raw.split("//")[0]
        if "#ASMSTART" in text:
            in_asm = True
        if "#ASMEND" in text:
            in_asm = False
        text = text.split(";")[0].strip()
        if not text:
            continue
        if text.endswith(":") and not text.startswith("."):
            pending.append(text[:-1])
            continue
        if re.match(r"^\.L[\w$]+:$", text):
            pending.append(text[:-1])
            continue
        if text.startswith("."):
            continue
        parts = text.split(None, 1)
        op = parts[0]
        if not re.match(r"^(v_|s_|ds_|global_|buffer_|flat_|scratch_)", op):
            continue
        ops = [t for t in re.split(r",\s*(?![^\[]*\])", parts[1])] if len(parts) > 1 else []
        i = Ins()
        i.op, i.line, i.text, i.asm = op, ln, text, in_asm
        stores = op.startswith(("ds_write", "ds_max", "ds_min", "ds_add", "global_store", "buffer_store", "flat_store", "scratch_store", "s_"))
        if op.startswith("v_cmp") or stores:
            i.dst, srcs = set(), ops
        else:
            i.dst, srcs = (regs(ops[0]) if ops else set()), ops[1:]
        if op.startswith("v_permlane") and "swap" in op:  # both operands are read and written
            i.dst = regs(ops[0]) | regs(ops[1])
            srcs = ops
        if op.startswith("v_mfma"):
            srcs = ops[1:]
        if op.startswith(("v_fmac", "v_mac", "v_dot")) or "_dpp" in text and False:
            srcs = ops
        i.src = set()
        for t in srcs:
            i.src |= regs(t.split()[0] if t.split() else t)
        for lb in pending:
            labels[lb] = len(ins)
        pending = []
        ins.append(i)
    return ins, labels


def predecessors(ins, labels):
    """index -> list of predecessor indices (fall-through and branch sources)."""
    target_of = {}
    for k, i in enumerate(ins):
        if i.op.startswith(("s_branch", "s_cbranch")):
            m = re.search(r"(\.L[\w$]+)", i.text)
            if m and m.group(1) in labels:
                target_of.setdefault(labels[m.group(1)], []).append(k)
    preds = {}
    for k in range(len(ins)):
        p = []
        if k > 0 and not ins[k - 1].op.startswith(("s_branch", "s_endpgm", "s_setpc")):
            p.append(k - 1)
        p += target_of.get(k, [])
        preds[k] = p
    return preds


def wait_states(i):
    if i.op == "s_nop":
        m = re.search(r"s_nop\s+(\d+)", i.text)
        return int(m.group(1)) + 1
    return 1


def producers(ins, preds, k, regset, limit):
    """Producers of any register of `regset` reachable backwards from instruction k within `limit` wait states:
    yields (producer, wait states between producer and consumer)."""
    out = []
    stack = [(p, 0, frozenset(regset)) for p in preds[k]]
    seen = set()
    while stack:
        j, ws, live = stack.pop()
        if (j, live) in seen and ws >= limit:
            continue
        seen.add((j, live))
        i = ins[j]
        hit = i.dst & live
        if hit and i.op != "s_nop":
            out.append((i, ws))
            live = live - hit
        ws2 = ws + wait_states(i)
        if live and ws2 < limit:
            for p in preds[j]:
                stack.append((p, ws2, live))
    return out


def lint(path):
    ins, labels = parse(path)
    preds = predecessors(ins, labels)
    findings = []
    for k, i in enumerate(ins):
        base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", i.op)
        where = f"{path}:{i.line}: {i.text}"
        if i.op.startswith("v_pk_") and re.search(r"_f32(_e64)?$", i.op):
            findings.append(f"{where}    packed-fp32 instruction")
        if base in CHECKED or "_dpp" in i.op or " quad_perm:" in i.text or " row_" in i.text:
            is_dpp = "_dpp" in i.op or "quad_perm:" in i.text or "row_" in i.text
            # inline asm may carry its own leading s_nop (MCP_MAX_DPP does): it was parsed as a separate instruction
            for prod, ws in producers(ins, preds, k, i.src, MAX_LOOKBACK):
                need = 0
                if prod.op.startswith("v_mfma"):
                    need = mfma_wait(prod.op)
                elif prod.op.startswith(TRANS):
                    need = 1
                if is_dpp and prod.op.startswith("v_"):
                    need = max(need, 2)
                if ws < need and (base in CHECKED or is_dpp and i.asm):
                    findings.append(f"{where}    reads the result of `{prod.text}` (line {prod.line}) after {ws} wait state(s), needs {need}")
        elif (i.op.startswith(("v_", "ds_", "global_", "buffer_", "flat_", "scratch_")) and not i.op.startswith("v_mfma")) and i.src:
            # (round 5) EVERY consumer of a matrix-core result held in ordinary VGPRs, not only the opcodes inline asm can hide: with
            # -mllvm -amdgpu-mfma-vgpr-form the accumulators are read by vector / LDS / memory instructions directly, and on one path
            # (a branch over the partial-tile masking of attention_small_kernel) this compiler left 5 wait states where 18 are needed
            for prod, ws in producers(ins, preds, k, i.src, MAX_LOOKBACK):
                if prod.op.startswith("v_mfma"):
                    need = mfma_wait(prod.op)
                    if ws < need:
                        findings.append(f"{where}    reads the result of `{prod.text}` (line {prod.line}) after {ws} wait state(s), needs {need}")
        if i.op.startswith(("v_readlane", "v_readfirstlane")) or (i.op.startswith("v_permlane") and "swap" in i.op):
            need = 2 if "swap" in i.op else 1
            for prod, ws in producers(ins, preds, k, i.src, 4):
                pbase = re.sub(r"_(e32|e64|dpp|sdwa)$", "", prod.op)
                if (pbase in CHECKED or prod.asm) and ws < need:
                    findings.append(f"{where}    reads the result of `{prod.text}` (line {prod.line}, possibly inline asm) after {ws} wait state(s), needs {need}")
        if i.op.startswith("ds_max") and "u64" in i.op:
            # every forward path to the next s_barrier must pass an s_waitcnt with an lgkmcnt field
            j, ok = k + 1, False
            while j < len(ins) and j < k + 40:
                if ins[j].op == "s_waitcnt" and "lgkmcnt" in ins[j].text:
                    ok = True
                    break
                if ins[j].op == "s_barrier":
                    break
                j += 1
            if not ok and j < len(ins) and ins[j].op == "s_barrier":
                findings.append(f"{where}    no s_waitcnt lgkmcnt between this LDS atomic and the s_barrier at line {ins[j].line}")
    return findings, len(ins)


def main():
    total, bad = 0, []
    for p in sys.argv[1:]:
        f, n = lint(p)
        total += n
        bad += f
    for line in bad:
        print(line)
    print(f"isa_lint: {total} instructions in {len(sys.argv) - 1} file(s), {len(bad)} finding(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

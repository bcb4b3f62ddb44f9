"""Audit of the emitted gfx950 assembly of the LDS-DMA ring kernels (conv_igemm / conv_wgrad / ...).

The ring's correctness rests on three properties of the INSTRUCTION STREAM that the HIP source cannot enforce
(ADVICE r01 #1):

  RAW  the counted `s_waitcnt vmcnt(N)` in front of the K-step barrier assumes that a wave issues exactly
       `dma_per_stage` vector-memory operations per loop iteration, all of them LDS-DMAs, in program order:
       a compiler-added VMEM op (a scratch spill, a hoisted global load) would shift the count and a stage would be
       read before it landed.
  WAR  a stage is overwritten by the DMAs issued right after the barrier of the NEXT iteration; every `ds_read` of an
       iteration must therefore have RETURNED (s_waitcnt lgkmcnt(0), or every result consumed) before the wave
       arrives at that barrier.  `s_barrier` waits for no counter.
  no scratch traffic inside a barrier loop.

Usage:  python tools/isa_audit.py [file.hip ...]      (compiles with -save-temps into a temp dir, prints a table,
exit code 1 on a violated property).  tests/test_host_cpu.py::test_ring_kernels_isa_audit runs the same check.
"""
import os
import re
import subprocess
import sys
import tempfile

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vfd_gan_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "--cuda-device-only"]

VMEM_RE = re.compile(r"^\s*(global_|buffer_|scratch_|flat_)(load|store|atomic)")
DMA_RE = re.compile(r"^\s*(global_load_lds_|buffer_load_.*\blds\b)")
DSREAD_RE = re.compile(r"^\s*ds_(read|load)")
DSWRITE_RE = re.compile(r"^\s*ds_(write|store)")
LABEL_RE = re.compile(r"^(\.LBB\d+_\d+):")
BRANCH_RE = re.compile(r"^\s*s_cbranch_\w+\s+(\.LBB\d+_\d+)|^\s*s_branch\s+(\.LBB\d+_\d+)")


def compile_to_asm(src, outdir):
    hipcc = "/opt/rocm/bin/hipcc"
    src = os.path.abspath(src)
    base = os.path.splitext(os.path.basename(src))[0]
    r = subprocess.run([hipcc, *FLAGS, "-save-temps", "-c", src, "-o", os.path.join(outdir, base + ".o")],
                       cwd=outdir, capture_output=True, text=True, timeout=1200)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-3000:])
    for f in os.listdir(outdir):
        if f.startswith(base + "-hip-amdgcn") and f.endswith(".s"):
            return os.path.join(outdir, f)
    raise RuntimeError("no device .s produced for %s" % src)


def split_kernels(text):
    """-> {mangled name: (lines, metadata dict)}"""
    out = {}
    lines = text.splitlines()
    names = [m.group(1) for m in (re.match(r"^\s*\.amdhsa_kernel\s+(\S+)", l) for l in lines) if m]
    meta = {}
    # YAML metadata at the end of the file: one "- .agpr_count: ..." block per kernel, keys in alphabetical order
    # (.name sits in the middle of its block), so collect a block and file it under its .name when the block ends
    block = {}

    def flush():
        if ".name" in block:
            meta[block[".name"]] = {k[1:]: v for k, v in block.items() if k != ".name"}
        block.clear()

    in_meta = False
    for l in lines:
        if "amdhsa.kernels:" in l:
            in_meta = True
            continue
        if not in_meta:
            continue
        if re.match(r"^\s*-\s+\.\w+:", l) and not re.match(r"^\s{4,}-\s", l):
            flush()
        m = re.match(r"^\s*-?\s*(\.name):\s+(\S+)", l)
        if m and not l.startswith("      "):
            block[".name"] = m.group(2)
        for key in (".vgpr_count", ".agpr_count", ".sgpr_count", ".vgpr_spill_count", ".sgpr_spill_count",
                    ".private_segment_fixed_size", ".group_segment_fixed_size"):
            m = re.match(r"^\s*-?\s*%s:\s+(\d+)" % re.escape(key), l)
            if m:
                block[key] = int(m.group(1))
    flush()
    for n in names:
        try:
            beg = next(i for i, l in enumerate(lines) if l.startswith(n + ":"))
        except StopIteration:
            continue
        end = next(i for i in range(beg, len(lines)) if "s_endpgm" in lines[i] and not lines[i].strip().startswith(";"))
        # the last s_endpgm of the function: continue to .Lfunc_end
        for j in range(end, len(lines)):
            if lines[j].startswith(".Lfunc_end"):
                end = j
                break
        out[n] = (lines[beg:end], meta.get(n, {}))
    return out


def demangle(names):
    try:
        r = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), capture_output=True, text=True)
        return dict(zip(names, r.stdout.splitlines()))
    except Exception:
        return {n: n for n in names}


def lgkm_zero(line):
    m = re.search(r"lgkmcnt\((\d+)\)", line)
    return bool(m) and int(m.group(1)) == 0 and "s_waitcnt" in line


def audit_kernel(lines):
    """Finds the natural loops (label ... backward branch to it) that contain an s_barrier and checks them."""
    code = [(i, l) for i, l in enumerate(lines) if l.strip() and not l.strip().startswith(";")]
    label_pos = {}
    for k, (_, l) in enumerate(code):
        m = LABEL_RE.match(l)
        if m:
            label_pos[m.group(1)] = k
    loops = []
    for k, (_, l) in enumerate(code):
        m = BRANCH_RE.match(l)
        if m:
            tgt = m.group(1) or m.group(2)
            if tgt in label_pos and label_pos[tgt] <= k:
                loops.append((label_pos[tgt], k))
    # innermost-first, keep loops with a barrier
    res = []
    for beg, end in sorted(set(loops), key=lambda t: t[1] - t[0]):
        body = [l for _, l in code[beg:end + 1]]
        raw_body = lines[code[beg][0]:code[end][0] + 1]       # with the ;;#ASMSTART / ;;#ASMEND markers
        if not any("s_barrier" in l for l in body):
            continue
        if any(b2 >= beg and e2 <= end and (b2, e2) != (beg, end) and any("s_barrier" in x for _, x in code[b2:e2 + 1])
               for b2, e2 in loops):
            continue   # an outer loop of a barrier loop: the inner one is the ring
        nbar = sum("s_barrier" in l for l in body)
        vmem = [l.strip() for l in body if VMEM_RE.match(l)]
        dma = [l for l in vmem if DMA_RE.match(l)]
        scratch = [l for l in vmem if l.startswith("scratch_")]
        # rotate so that the body starts right after the (first) barrier, then look at what follows the last ds_read
        b0 = next(i for i, l in enumerate(body) if "s_barrier" in l)
        rot = body[b0 + 1:] + body[:b0 + 1]
        last_read = max((i for i, l in enumerate(rot) if DSREAD_RE.match(l)), default=None)
        war_ok = True
        if last_read is not None:
            # between the last LDS read and the next barrier there must be a full lgkmcnt(0) wait
            nxt_bar = next(i for i in range(last_read, len(rot)) if "s_barrier" in rot[i])
            war_ok = any(lgkm_zero(l) for l in rot[last_read + 1:nxt_bar + 1])
        vm_waits = [int(m.group(1)) for l in body for m in [re.search(r"s_waitcnt.*vmcnt\((\d+)\)", l)] if m]
        # waits the COMPILER put into the loop (outside ;;#ASMSTART / ;;#ASMEND): with the DMAs issued from inline asm it
        # tracks no LDS-DMA, so any vmcnt wait of its own inside a ring loop means an unexpected VMEM dependency
        in_asm, own_waits = False, []
        for l in raw_body:
            t = l.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
            elif t.startswith(";;#ASMEND"):
                in_asm = False
            elif not in_asm and re.search(r"s_waitcnt.*vmcnt\(", t):
                own_waits.append(t)
        res.append(dict(barriers=nbar, vmem=len(vmem), dma=len(dma), other_vmem=[l for l in vmem if l not in dma][:4],
                        scratch=len(scratch), ds_reads=sum(bool(DSREAD_RE.match(l)) for l in body),
                        ds_writes=sum(bool(DSWRITE_RE.match(l)) for l in body),
                        mfma=sum("v_mfma" in l for l in body), war_ok=war_ok, vmcnt_waits=vm_waits, compiler_waits=own_waits))
    return res


def audit_file(src, outdir):
    asm = compile_to_asm(src, outdir)
    with open(asm) as f:
        kernels = split_kernels(f.read())
    names = demangle(list(kernels))
    report = []
    for n, (lines, meta) in kernels.items():
        loops = audit_kernel(lines)
        viol = []
        if meta.get("vgpr_spill_count", 0) or meta.get("private_segment_fixed_size", 0):
            viol.append("spill: %d VGPRs, %d B scratch" % (meta.get("vgpr_spill_count", 0), meta.get("private_segment_fixed_size", 0)))
        for lp in loops:
            if lp["dma"] == 0:
                continue    # a barrier loop without LDS-DMA (reductions): __syncthreads semantics, not a ring
            if lp["scratch"]:
                viol.append("scratch traffic inside the ring loop")
            if lp["vmem"] != lp["dma"]:
                viol.append("non-DMA VMEM ops inside the ring loop: %r" % lp["other_vmem"])
            if not lp["war_ok"]:
                viol.append("WAR: no lgkmcnt(0) between the last ds_read and the barrier")
            # the counted wait must be a multiple of the per-iteration DMA count (STAGES-2 stages in flight)
            if lp["compiler_waits"]:
                viol.append("compiler-inserted vmcnt wait inside the ring loop: %r" % lp["compiler_waits"][:2])
        # kernels that issue their DMAs from inline asm write M0 there; the compiler must have no M0 use of its own
        in_asm, asm_dma, own_m0 = False, 0, []
        for l in lines:
            t = l.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
            elif t.startswith(";;#ASMEND"):
                in_asm = False
            elif in_asm and DMA_RE.match(l):
                asm_dma += 1
            elif not in_asm and not t.startswith(";") and re.search(r"\bm0\b", t):
                own_m0.append(t)
        if asm_dma and own_m0:
            viol.append("compiler-generated M0 use beside asm LDS-DMA: %r" % own_m0[:3])
        report.append(dict(name=names.get(n, n), meta=meta, loops=loops, violations=viol))
    return report


def main(argv):
    files = argv or [os.path.join(CSRC, f) for f in ("conv_igemm.hip", "conv_halo.hip", "conv_wgrad.hip", "conv_wgrad_halo.hip", "conv_small.hip")]
    bad = 0
    with tempfile.TemporaryDirectory() as td:
        for f in files:
            for r in audit_file(f, td):
                ring = [lp for lp in r["loops"] if lp["dma"]]
                if not ring and not r["violations"]:
                    continue
                m = r["meta"]
                short = re.sub(r"\(anonymous namespace\)::", "", r["name"])[:90]
                print("%-92s vgpr %3d agpr %3d lds %6d" % (short, m.get("vgpr_count", -1), m.get("agpr_count", -1),
                                                            m.get("group_segment_fixed_size", -1)))
                for lp in ring:
                    print("    ring loop: %d barrier, %d DMA / iteration, vmcnt waits %s, %d ds_read, %d mfma, WAR %s" %
                          (lp["barriers"], lp["dma"], lp["vmcnt_waits"], lp["ds_reads"], lp["mfma"], "ok" if lp["war_ok"] else "VIOLATED"))
                for v in r["violations"]:
                    bad += 1
                    print("    !! " + v)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))

"""hipcc driver for every gfx950 artefact of this package (libndq.so and the generated kernels), with one extra step
between code generation and assembly: a fix-up pass over the device assembly.

Why: on gfx950 (MI355X, ROCm 7.2 / clang 22) a packed-fp32 VALU instruction (``v_pk_mul_f32`` / ``v_pk_fma_f32`` / ...)
that is IMMEDIATELY followed by a bf16 MFMA can deliver a wrong low-half result in lanes 48..63 (the last quarter of the
wave): the lanes get the value of the preceding packed instruction's operand.  Found as a non-deterministic dW1 entry of
one closure kernel, bisected at the assembly level, and reproduced in isolation by ``neurodiffeq_amd/csrc/canary_pk_war.hip``
(``profiles/archive/r01/r01u_pk_mfma_hazard.txt``: ~11 % of the executions wrong; one wait state -- ``s_nop 0`` or any other
instruction -- between the two makes it exact; scalar ``v_mul_f32`` instead of the packed form is exact).  The compiler's
hazard recogniser does not know the pair, so ``fix_pk_mfma`` inserts the wait state itself: ~20 sites per kernel, one
cycle each.

Pipeline (what ``hipcc -shared`` does internally, split so that the assembly can be edited):
  1. hipcc --cuda-device-only -S          -> device assembly
  2. fix_pk_mfma
  3. clang (assembler) -> lld -> clang-offload-bundler   -> fat binary
  4. hipcc --cuda-host-only -fcuda-include-gpubinary     -> shared library
"""
import os
import re
import shutil
import subprocess

HIPCC = os.environ.get("NDQ_HIPCC", "/opt/rocm/bin/hipcc")
LLVM_BIN = os.environ.get("NDQ_LLVM_BIN", os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(HIPCC))),
                                                       "lib", "llvm", "bin"))
ARCH = "gfx950"
BASE_FLAGS = ["-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")]
# part of every cache key: bump when the fix-up rules change so that cached kernels are rebuilt
FIXUP_VERSION = "pk-opsel-scalar+mfma-nop-3"

_PK = re.compile(r"^\s*v_pk_\w+")
_MFMA = re.compile(r"^\s*v_s?mfmac?_\w+")


def _is_instruction(line):
    s = line.strip()
    return bool(s) and not s.startswith((";", "//", ".")) and not s.endswith(":")


def fix_pk_mfma(asm_text, rule="mfma"):
    """Insert ``s_nop 0`` between a packed VALU instruction and an MFMA that directly follows it (labels, comments and
    directives in between do not separate them at run time).  Returns (patched text, number of sites).
    ``rule`` (experiments, -DNDQ_FIXUP_RULE=<name> among the extra flags): "mfma" (the shipped rule), "all" (after every
    packed op), "mem" (additionally before a directly following scratch / global / LDS instruction), "before" (a wait
    state in front of every packed op as well)."""
    out, sites, prev_pk = [], 0, False
    for line in asm_text.split("\n"):
        if _is_instruction(line):
            s = line.strip()
            follows = bool(_MFMA.match(line)) or rule == "all" or \
                (rule == "mem" and s.startswith(("scratch_", "global_", "ds_", "buffer_", "flat_")))
            if prev_pk and follows and not s.startswith("s_nop"):
                out.append("\ts_nop 0")
                sites += 1
            prev_pk = bool(_PK.match(line))
            if prev_pk and rule == "before":
                out.append("\ts_nop 0")
                sites += 1
        out.append(line)
    return "\n".join(out), sites


_PK_F32 = re.compile(r"^(\s*)v_pk_(mul|add|fma)_f32\s+(.*)$")
_MOD = re.compile(r"\b(op_sel|op_sel_hi|neg_lo|neg_hi):\[([01,]+)\]")


def _half(operand, hi):
    """Register / constant that the low (hi = 0) or high (hi = 1) half of a packed operand names."""
    m = re.fullmatch(r"([vs])\[(\d+):(\d+)\]", operand)
    if m:
        return f"{m.group(1)}{int(m.group(2)) + hi}"
    return operand                                   # inline constant / literal: the same value in both halves


def scalarize_pk(asm_text, which="opsel"):
    """Rewrite packed-fp32 VALU instructions as two scalar VOP3 instructions (experiments / hazard work-arounds).
    ``which``: "opsel" -- only instructions whose LOW half selects a high source half (op_sel with a 1 in it);
    "opsel_hi" -- additionally those whose high half selects the low half of a REGISTER pair; "all" -- every one.
    Instructions whose halves would clobber each other's sources are left alone.  Returns (text, rewritten, skipped)."""
    out, done, skipped = [], 0, 0
    for line in asm_text.split("\n"):
        m = _PK_F32.match(line)
        if not m:
            out.append(line)
            continue
        indent, op, rest = m.groups()
        rest = rest.split(";")[0].strip()
        mods = {k: [int(x) for x in v.split(",")] for k, v in _MOD.findall(rest)}
        ops = [o.strip() for o in _MOD.sub("", rest).strip().rstrip(",").split(",")]
        ops = [o for o in ops if o]
        nsrc = 3 if op == "fma" else 2
        if len(ops) != nsrc + 1 or not re.fullmatch(r"v\[\d+:\d+\]", ops[0]):
            out.append(line)
            continue
        sel = mods.get("op_sel", [0] * nsrc) + [0] * nsrc
        sel_hi = mods.get("op_sel_hi", [1] * nsrc) + [1] * nsrc
        neg_lo = mods.get("neg_lo", [0] * nsrc) + [0] * nsrc
        neg_hi = mods.get("neg_hi", [0] * nsrc) + [0] * nsrc
        is_pair = [bool(re.fullmatch(r"[vs]\[\d+:\d+\]", o)) for o in ops[1:]]
        # anything that is neither a 64-bit register pair nor a plain numeric constant (clamp / omod suffixes, special
        # registers, symbols ...): not an instruction this rewrite understands -- leave it alone
        if not all(p or re.fullmatch(r"-?(\d+(\.\d+)?|0x[0-9a-fA-F]+)", o) for p, o in zip(is_pair, ops[1:])):
            out.append(line)
            skipped += 1
            continue
        has_opsel = any(sel[i] for i in range(nsrc))
        has_hi = any(is_pair[i] and not sel_hi[i] for i in range(nsrc))
        if not (which == "all" or (which == "opsel" and has_opsel) or (which == "opsel_hi" and (has_opsel or has_hi))):
            out.append(line)
            continue
        d_lo, d_hi = _half(ops[0], 0), _half(ops[0], 1)
        src_lo = [("-" if neg_lo[i] else "") + _half(ops[1 + i], sel[i] if is_pair[i] else 0) for i in range(nsrc)]
        src_hi = [("-" if neg_hi[i] else "") + _half(ops[1 + i], sel_hi[i] if is_pair[i] else 0) for i in range(nsrc)]
        mnem = {"mul": "v_mul_f32_e64", "add": "v_add_f32_e64", "fma": "v_fma_f32"}[op]
        lo = f"{indent}{mnem} {d_lo}, " + ", ".join(src_lo)
        hi = f"{indent}{mnem} {d_hi}, " + ", ".join(src_hi)
        lo_clobbers_hi = any(x.lstrip("-") == d_lo for x in src_hi)
        hi_clobbers_lo = any(x.lstrip("-") == d_hi for x in src_lo)
        if lo_clobbers_hi and hi_clobbers_lo:
            # each half would overwrite a source of the other.  The case the compiler emits is the pair sum / product
            # "x.lo (op) x.hi in both halves" (v_pk_add_f32 v[a:a+1], v[a:a+1], v[a:a+1] op_sel:[0,1] op_sel_hi:[1,0]): both
            # halves compute the SAME value from the same operands -- compute it once, copy it.  Anything else is left
            # alone here and stopped by verify_fixup (the build fails closed).
            commutes = sorted(src_lo[:2]) == sorted(src_hi[:2]) and (nsrc == 2 or src_lo[2] == src_hi[2])
            if commutes:
                out += [lo, f"{indent}v_mov_b32_e32 {d_hi}, {d_lo}"]
                done += 1
                continue
            # general case: exchange the two destination registers first (v_swap_b32), rename them in the sources; the
            # halves then stop reading each other's destination in at least one order
            def ren(x):
                neg, r = ("-", x[1:]) if x.startswith("-") else ("", x)
                return neg + (d_hi if r == d_lo else d_lo if r == d_hi else r)
            s_lo, s_hi = [ren(x) for x in src_lo], [ren(x) for x in src_hi]
            lo2 = f"{indent}{mnem} {d_lo}, " + ", ".join(s_lo)
            hi2 = f"{indent}{mnem} {d_hi}, " + ", ".join(s_hi)
            lo_first_ok = not any(x.lstrip("-") == d_lo for x in s_hi)
            hi_first_ok = not any(x.lstrip("-") == d_hi for x in s_lo)
            if lo_first_ok or hi_first_ok:
                out += [f"{indent}v_swap_b32 {d_lo}, {d_hi}"] + ([lo2, hi2] if lo_first_ok else [hi2, lo2])
                done += 1
                continue
            out.append(line)
            skipped += 1
            continue
        out += [hi, lo] if lo_clobbers_hi else [lo, hi]
        done += 1
    return "\n".join(out), done, skipped


_PK_ANY_F32 = re.compile(r"^\s*v_pk_\w+_f32\s")
STATS = {}            # output path -> dict(split=, left_alone=, nops=) of the last build through this process


def verify_fixup(asm_text):
    """Post-condition of the fix-up pass, checked on the assembly that is about to be assembled -- the pass FAILS CLOSED:
    (1) no packed-fp32 instruction whose low half selects a high source half (``op_sel`` with a 1) may be left: either the
    rewrite understood it, or the build stops (an encoding the ROCm 7.2 code generator did not emit when the rules were
    written -- e.g. after a compiler update -- must be looked at, not waved through);
    (2) no packed VALU instruction may be directly followed by an MFMA.  Raises RuntimeError."""
    prev_pk = None
    # One kind of instruction has no two-instruction rewrite: a genuine two-in / two-out update, (L, H) <- (a L + H, b H + L),
    # where each half reads BOTH destination registers (it would need a spare register).  What is known about the op_sel
    # forms: the direct hazard -- such an instruction immediately before a bf16 MFMA -- is removed by the
    # wait state fix_pk_mfma inserts; the second, unexplained failure was only ever seen in kernels that SPILL.  So such an
    # instruction may stay in a module without scratch, and stops the build in a module with scratch.
    spills = any(int(v) > 0 for v in re.findall(r"^; ScratchSize: (\d+)", asm_text, flags=re.M))
    for ln, line in enumerate(asm_text.split("\n"), 1):
        if not _is_instruction(line):
            continue
        s = line.strip()
        if _PK_ANY_F32.match(line):
            m = re.search(r"\bop_sel:\[([01,]+)\]", s)
            if m and "1" in m.group(1):
                two_in_two_out = False
                pm = _PK_F32.match(line)
                if pm:
                    ops = [o.strip() for o in _MOD.sub("", pm.group(3).split(";")[0]).strip().rstrip(",").split(",") if o.strip()]
                    two_in_two_out = len(ops) >= 3 and ops[1:].count(ops[0]) >= 2
                if not (two_in_two_out and not spills):
                    raise RuntimeError(f"assembly fix-up: packed fp32 instruction with op_sel left in the output (line {ln}): {s}\n"
                                       "-- an encoding _hipcc.scalarize_pk has no rewrite for"
                                       + (" in a kernel module that spills registers" if two_in_two_out else "")
                                       + "; NDQ_NO_PK_MFMA_FIX=1 builds without the pass (and without its protection)")
        if prev_pk is not None and _MFMA.match(line):
            raise RuntimeError(f"assembly fix-up: packed VALU instruction directly followed by an MFMA (line {ln}): {prev_pk} / {s}")
        prev_pk = s if _PK.match(line) else None


def fixup_enabled():
    return os.environ.get("NDQ_NO_PK_MFMA_FIX", "0") != "1"


def _run(cmd, what):
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError(f"{what} failed:\n{' '.join(cmd)}\n{proc.stderr[-6000:]}")


_POOL = None          # (executor, [futures]) while a ``deferred`` block is active
_JOBS = int(os.environ.get("NDQ_BUILD_JOBS", 0)) or (os.cpu_count() or 4)


class deferred:
    """``with _hipcc.deferred():`` -- inside the block, ``compile_shared(..., defer=True)`` only QUEUES the build on a pool
    of NDQ_BUILD_JOBS (default: all cores) worker threads (the work happens in hipcc / clang subprocesses) and returns
    at once; leaving the block waits for every queued build and raises the first failure.  ``__graft_entry__.build`` and
    the first-use builds of a system's kernels (4-wave / 8-wave closure kernel, pointwise kernel) use it, so a full
    build costs about (total compile time) / cores."""

    def __enter__(self):
        global _POOL
        from concurrent.futures import ThreadPoolExecutor
        self.prev = _POOL
        _POOL = (ThreadPoolExecutor(max_workers=_JOBS), [], set())
        return self

    def __exit__(self, exc_type, exc, tb):
        global _POOL
        pool, futures, _ = _POOL
        _POOL = self.prev
        errors = []
        for f in futures:
            try:
                f.result()
            except Exception as e:      # noqa: BLE001 -- collected, the first one is re-raised below
                errors.append(e)
        pool.shutdown()
        if errors and exc_type is None:
            raise errors[0]
        return False


def compile_shared(sources, out, extra_flags=(), verbose=False, defer=False):
    """Build ``out`` (a shared library holding host code + the gfx950 code object) from HIP sources.  Returns the number
    of pk->mfma sites the fix-up pass separated.  Atomic: ``out`` is replaced only when everything succeeded.
    ``defer``: inside a ``deferred`` block, queue the build instead of running it (returns None)."""
    if defer and _POOL is not None:
        if out not in _POOL[2]:             # (several systems may share one generated kernel)
            _POOL[2].add(out)
            _POOL[1].append(_POOL[0].submit(compile_shared, sources, out, tuple(extra_flags), verbose))
        return None
    sources = [sources] if isinstance(sources, str) else list(sources)
    flags = BASE_FLAGS + list(extra_flags)
    import threading
    work = f"{out}.build{os.getpid()}_{threading.get_ident()}"
    os.makedirs(work, exist_ok=True)
    try:
        if not fixup_enabled():                       # experiments: the compiler's own output, one step
            tmp = os.path.join(work, "out.so")
            _run([HIPCC, f"--offload-arch={ARCH}"] + flags + ["-shared"] + sources + ["-o", tmp], "hipcc")
            os.replace(tmp, out)
            return 0
        sites, objs = 0, []
        STATS.pop(out, None)
        for i, src in enumerate(sources):
            asm = os.path.join(work, f"dev{i}.s")
            _run([HIPCC, f"--offload-arch={ARCH}", "--cuda-device-only", "-S"] + flags + [src, "-o", asm],
                 "hipcc (device code generation)")
            rule = next((f.split("=", 1)[1] for f in flags if f.startswith("-DNDQ_FIXUP_RULE=")), "mfma")
            with open(asm) as fh:
                text = fh.read()
            split = next((f.split("=", 1)[1] for f in flags if f.startswith("-DNDQ_FIXUP_SPLIT=")), "opsel")
            if split != "none":
                text, n_split, n_skip = scalarize_pk(text, split)
                if verbose:
                    print(f"scalarized {n_split} packed op(s) ({split}), left {n_skip} alone", flush=True)
            text, n = fix_pk_mfma(text, rule)
            sites += n
            if split == "opsel" and rule == "mfma":       # (experiment variants of the rules are not held to the post-condition)
                verify_fixup(text)
            st = STATS.setdefault(out, dict(split=0, left_alone=0, nops=0))
            st["nops"] += n
            if split != "none":
                st["split"] += n_split
                st["left_alone"] += n_skip
            with open(asm, "w") as fh:
                fh.write(text)
            obj, co, fb = (os.path.join(work, f"dev{i}.{e}") for e in ("o", "out", "hipfb"))
            _run([os.path.join(LLVM_BIN, "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", f"-mcpu={ARCH}",
                  "-c", asm, "-o", obj], "assembling the device code")
            _run([os.path.join(LLVM_BIN, "lld"), "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared",
                  obj, "-o", co], "linking the device code")
            _run([os.path.join(LLVM_BIN, "clang-offload-bundler"), "-type=o", "-bundle-align=4096",
                  f"-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--{ARCH}", "-input=/dev/null",
                  f"-input={co}", f"-output={fb}"], "bundling the device code")
            host = os.path.join(work, f"host{i}.o")
            _run([HIPCC, f"--offload-arch={ARCH}", "--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", fb]
                 + flags + ["-c", src, "-o", host], "hipcc (host code)")
            objs.append(host)
        tmp = os.path.join(work, "out.so")
        _run([HIPCC, "-shared", "-fPIC"] + objs + ["-o", tmp], "linking the shared library")
        os.replace(tmp, out)
        if verbose:
            print(f"built {out}: {sites} pk->mfma site(s) separated", flush=True)
        return sites
    finally:
        shutil.rmtree(work, ignore_errors=True)

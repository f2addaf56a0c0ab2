#!/bin/bash
# Register / scratch / instruction-mix statistics of every kernel in one HIP source, as the shipped pipeline compiles it
# (device assembly only; no GPU needed).  usage: scripts/isa_stats.sh file.hip [extra hipcc flags...]
src=$1; shift
out=${ISA_OUT:-/tmp/isa_$$.s}
/opt/rocm/bin/hipcc --offload-arch=gfx950 --cuda-device-only -S -O3 -std=c++17 "$@" "$src" -o "$out" || exit 1
python3 - "$out" <<'P'
import re, sys
text = open(sys.argv[1]).read()
# per-kernel metadata
for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)(?=\n  - \.|\namdhsa\.target|\Z)", text, re.S):
    pass
names = re.findall(r"^\s+\.name:\s+(\S+)\s*$", text, re.M)
md = text[text.rfind(".amdgpu_metadata"):]
for blk in md.split("  - .agpr_count:")[1:]:
    blk = ".agpr_count:" + blk
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
    nm = g("name")
    print(f"{nm[:90]:90s} vgpr={g('vgpr_count')} agpr={g('agpr_count')} sgpr={g('sgpr_count')} "
          f"spill={g('vgpr_spill_count')} scratch={g('private_segment_fixed_size')} lds={g('group_segment_fixed_size')}")
# instruction mix per function body
for fn in re.finditer(r"^(\w+):[^\n]*\n(.*?)^\s*s_endpgm", text, re.S | re.M):
    body = fn.group(2)
    ins = [l.split()[0] for l in body.split("\n") if l.strip() and not l.strip().startswith((";", ".", "//")) and not l.strip().endswith(":")]
    c = lambda pat: sum(1 for i in ins if re.match(pat, i))
    print(f"{fn.group(1)[:60]:60s} insts={len(ins)} valu={c(r'v_(?!mfma|smfma)')} mfma_bf16={c(r'v_mfma.*bf16')} "
          f"mfma_f32={c(r'v_mfma_f32_16x16x4')} ds={c(r'ds_')} scratch={c(r'scratch_')} global={c(r'global_|buffer_')} "
          f"accvgpr={c(r'v_accvgpr')} s_nop={c(r's_nop')} waitcnt={c(r's_waitcnt')}")
P

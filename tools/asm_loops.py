#!/usr/bin/env python3
"""Innermost loops of one kernel in a gfx950 assembly listing (hipcc --cuda-device-only -S): instruction mix per loop.
usage: tools/asm_loops.py <file.s | source.hip> <mangled-name-substring> [max-len]
Given a .hip source it is compiled first (the build's flags) into /tmp/svr_asm/<stem>.s.
A loop = a backward branch to a label; reported: lines, VALU, of which v_mov, SALU, VMEM, LDS, and whether it holds a load."""
import re
import subprocess
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def assemble(src: Path) -> Path:
    from sunvolumerender_amd import _build

    out = Path("/tmp/svr_asm")
    out.mkdir(exist_ok=True)
    s = out / (src.stem + ".s")
    if not s.exists() or s.stat().st_mtime < max(p.stat().st_mtime for p in _build.CSRC.glob("*")):
        flags = _build.HIPCC_FLAGS if src.name not in _build.FAST_SOURCES else [f for f in _build.HIPCC_FLAGS if f not in _build.CONTRACT_FLAGS] + _build.FAST_FLAGS
        subprocess.run([_build._hipcc(), *flags, "--cuda-device-only", "-S", str(src), "-o", str(s)], check=True, stderr=subprocess.DEVNULL)
    return s


def main():
    path = Path(sys.argv[1])
    if path.suffix == ".hip":
        from sunvolumerender_amd import _build

        path = assemble(path if path.exists() else _build.CSRC / path.name)
    key = sys.argv[2]
    max_len = int(sys.argv[3]) if len(sys.argv) > 3 else 140
    lines = path.read_text().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and key in l)
    end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    labels = {}
    insts = []          # (text)
    for l in body:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."):
            continue
        insts.append(t.split(";")[0].strip())
    print(f"{lines[start].split(':')[0]}: {len(insts)} instructions")
    loops = []
    for i, t in enumerate(insts):
        m = re.match(r"s_cbranch\S*\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", t)
        if m:
            tgt = labels.get(m.group(1) or m.group(2))
            if tgt is not None and tgt <= i and i - tgt <= max_len:
                loops.append((tgt, i))
    # innermost only
    inner = [lp for lp in loops if not any(o != lp and lp[0] <= o[0] and o[1] <= lp[1] for o in loops)]
    for a, b in inner:
        seg = insts[a:b + 1]
        valu = [x for x in seg if x.startswith("v_") and not x.startswith("v_readfirstlane")]
        mov = [x for x in valu if x.startswith("v_mov_b32") or x.startswith("v_accvgpr")]
        salu = [x for x in seg if x.startswith("s_") and not x.startswith("s_waitcnt") and not x.startswith("s_nop")]
        vmem = [x for x in seg if x.startswith(("global_", "buffer_", "flat_", "scratch_"))]
        lds = [x for x in seg if x.startswith("ds_")]
        print(f"  loop @{a:6d}..{b:6d}: {len(seg):4d} instr  VALU {len(valu):3d} (v_mov {len(mov):2d})  SALU {len(salu):3d}  VMEM {len(vmem):2d}  LDS {len(lds):2d}")
        if "--dump" in sys.argv:
            for x in seg:
                print("      ", x)


if __name__ == "__main__":
    main()

"""Print per-kernel register / scratch usage from an AMDGPU assembly file (hipcc -S --cuda-device-only)."""
import re, sys
txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in txt.split("  - .agpr_count:")[1:]:
    d = dict(re.findall(r"\.(\w+):\s+(\S+)", blk.split("  - .agpr_count:")[0]))
    name = d.get("name", "?")
    if pat in name:
        print(f"{name[:90]:90s} vgpr={d.get('vgpr_count')} agpr={blk.split()[0]} sgpr={d.get('sgpr_count')} "
              f"vspill={d.get('vgpr_spill_count')} scratch={d.get('private_segment_fixed_size')} lds={d.get('group_segment_fixed_size')}")

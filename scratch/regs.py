import re, sys
s = open(sys.argv[1]).read()
meta = s[s.index("amdhsa.kernels:"):]
for blk in meta.split("  - .agpr_count")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    if not re.search(r"k_extend|k_tail|k_shade", name): continue
    g = lambda k: re.search(r"\.%s:\s+(\d+)" % k, blk).group(1)
    short = re.sub(r"^_Z\d+", "", name)[:40]
    print(f"{short:42s} vgpr {g('vgpr_count'):>3} spill {g('vgpr_spill_count'):>2} sgpr {g('sgpr_count'):>3} lds {g('group_segment_fixed_size'):>6} scratch {g('private_segment_fixed_size'):>4}")

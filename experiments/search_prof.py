"""Design check: instruction counters of one search launch (k = 256)."""
from basebandboard_amd import gf2
gf2.search(256, seed=1, first=0, count=256)
idx, _, st = gf2.search(256, seed=1, first=1 << 32, count=1 << 17)
print(idx, st)

import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import bench_kernels as bk
for B in (32, 11, 10):
    bk.attn_case(B, 6, 1370)

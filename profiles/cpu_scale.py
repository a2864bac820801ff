import sys, time, os
sys.path.insert(0, '.')
import __graft_entry__ as g
import numpy as np
o = g.load_oracle()
print("affinity", len(os.sched_getaffinity(0)))
for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpuset.cpus.effective"):
    try: print(path, open(path).read().strip())
    except Exception as e: print(path, "n/a")
os.system("lscpu | egrep 'Model name|Socket|NUMA node|Thread|Core' | head -12")
for th in (1, 8, 16, 32, 64, 128, 256):
    t0 = time.time()
    ms, fill = o.bench(np.float64, 4096, 60, 512, th, 4)
    print("4096x60x512 threads", th, "ms", [round(x,1) for x in ms], "Mcells/s", round(4096*60*512/np.median(ms[1:])/1e3,1), "fill", round(fill,2), "wall", round(time.time()-t0,1), flush=True)

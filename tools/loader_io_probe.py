"""tools/loader_io_probe.py (GPU box): where the time of a C2-size lm_index_open goes when the seed passes are done in half of it -
the read rate of the saved genome batch files against that of the seed chunk files, as the OS serves them right after
lm_index_save wrote them (page cache or disk), with 1 and 8 reader threads."""
import os, sys, time, tempfile, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lexicmap_amd as la

def read_rate(paths, nthreads):
    total = sum(os.path.getsize(p) for p in paths)
    jobs = []
    for p in paths:
        n = os.path.getsize(p)
        for o in range(0, n, 64 << 20):
            jobs.append((p, o, min(64 << 20, n - o)))
    lock = threading.Lock()
    def body():
        fds = {}
        while True:
            with lock:
                if not jobs:
                    break
                p, o, ln = jobs.pop()
            fd = fds.get(p) or os.open(p, os.O_RDONLY)
            fds[p] = fd
            left = ln
            while left:
                b = os.pread(fd, min(left, 8 << 20), o + ln - left)
                left -= len(b)
        for fd in fds.values():
            os.close(fd)
    t0 = time.time()
    th = [threading.Thread(target=body) for _ in range(nthreads)]
    [t.start() for t in th]; [t.join() for t in th]
    dt = time.time() - t0
    return total, dt

gi = la.Index.synthetic(10000, 5_000_000, 100, seed=1000, max_div=0.10)
d = os.path.join(tempfile.mkdtemp(prefix="lm_probe_"), "saved.lmi")
t0 = time.time(); gi.save(d, chunks=32); print("save %.1f s" % (time.time() - t0)); gi.close()
gen = [os.path.join(r, f) for r, _, fs in os.walk(os.path.join(d, "genomes")) for f in fs if f == "genomes.bin"]
seeds = [os.path.join(d, "seeds", f) for f in os.listdir(os.path.join(d, "seeds")) if f.endswith(".bin")]
print("genome batch files:", len(gen), "seed chunk files:", len(seeds))
print(open("/proc/meminfo").read().split("\n")[0:5])
for label, paths in (("genomes", gen), ("seeds", seeds), ("genomes again", gen)):
    for nt in (8, 1):
        tot, dt = read_rate(paths, nt)
        print("%-14s %2d threads: %.2f GB in %.2f s = %.2f GB/s" % (label, nt, tot / 1e9, dt, tot / 1e9 / dt))
t0 = time.time(); g2 = la.Index(d); print("open %.2f s" % (time.time() - t0)); g2.close()

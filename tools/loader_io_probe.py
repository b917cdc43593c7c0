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
# the host side of the genome reader alone (lm_format.cpp: load_index + load_index_genomes), no HIP: into the host store, and
# with a sink that only counts - beside nothing, and beside 16 threads decoding the seed files as the loader's do
import subprocess
HARNESS = r"""
#include <chrono>
#include <cstdio>
#include <future>
#include <vector>
#include "lm_format.h"
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    for (int mode = 0; mode < 3; mode++) {
        lm::HostIndex h;
        int st = 0;
        std::string e = lm::load_index(argv[1], 0, 1, h, st, false);
        if (!e.empty()) { printf("load_index: %s\n", e.c_str()); return 1; }
        long long bytes = 0;
        if (mode >= 1) h.gbits_sink = [&](const uint8_t *, size_t n, int64_t) { bytes += (long long)n; return true; };
        std::vector<std::future<void>> dec;
        std::vector<lm::SeedChunk> slots(16);
        double t0 = now();
        if (mode == 2)
            for (int t = 0; t < 16; t++)
                dec.push_back(std::async(std::launch::async, [&, t]() {
                    for (size_t i = (size_t)t; i < h.seed_files.size(); i += 16) { int s2 = 0, ap = -1; lm::decode_seed_chunk(h.seed_files[i], h, slots[(size_t)t], s2, ap); }
                }));
        e = lm::load_index_genomes(argv[1], h, st);
        double t1 = now();
        for (auto &f : dec) f.get();
        printf("mode %d (%s): load_index_genomes %.2f s (%s), %lld sink bytes, %zu store bytes, decoders done after %.2f s\n", mode,
               mode == 0 ? "host store" : mode == 1 ? "counting sink" : "counting sink beside 16 decoders", t1 - t0, e.empty() ? "ok" : e.c_str(), bytes, h.gbits.size(), now() - t0);
    }
    return 0;
}
"""
csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lexicmap_amd", "csrc")
src = os.path.join(tempfile.mkdtemp(), "probe.cpp")
open(src, "w").write(HARNESS)
exe = src[:-4]
subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-I", csrc, "-o", exe, src, os.path.join(csrc, "lm_format.cpp")])
print(subprocess.run([exe, d], capture_output=True, text=True).stdout)
t0 = time.time(); g2 = la.Index(d); print("open %.2f s" % (time.time() - t0)); g2.close()

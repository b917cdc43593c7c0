#!/usr/bin/env python3
"""adopt_arena_reserve.py [--root DIR]: LM_ARENA_RESERVE_PCT (default 0 = off) - the scratch arena of a handle takes ONE slab of
that share of the scratch budget from the device at the first search; the phase buffers of both lanes are then carved from it
and the device is not asked again while it suffices.  Why: hipMalloc / hipFree synchronise the device, so slabs cut on demand
make one lane's allocation wait for the other lane's persistent WFA kernels - a fresh two-lane handle took three C3 steps to
settle (17.5, 14.5, 12.2 s, profiles/r04_c3_steady.json).  The smallest form of experiments/lane_arena (which separates the
lanes' slabs); off by default, so the product does not change until a GPU run says what share works
(LM_ARENA_RESERVE_PCT=60 .. 80 on a fresh C3 handle: the first step should cost what the fourth does).  Asserted edits;
tests/test_adopt_scripts_cpu.py applies it to a copy of the tree and runs the reserved arena over a fake device."""
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) >= 3 and sys.argv[1] == "--root":
    root = os.path.abspath(sys.argv[2])
csrc = os.path.join(root, "lexicmap_amd", "csrc")


def edit(path, pairs):
    s = open(path).read()
    for old, new in pairs:
        assert s.count(old) == 1, (path, old[:70], s.count(old))
        s = s.replace(old, new)
    open(path, "w").write(s)


edit(os.path.join(csrc, "lm_internal.h"), [
    ("    void trim_locked() { // hand the slabs without a live block back to the device\n",
     "    // One slab of `bytes` from the device NOW: what is carved later comes out of it while it suffices (best fit over all\n"
     "    // slabs, as before).  false: the device refused, nothing changes.  An empty reserved slab goes back to the device like\n"
     "    // any other when an allocation fails (trim_locked).\n"
     "    bool reserve(size_t bytes) {\n"
     "        bytes = bytes / ALIGN * ALIGN;\n"
     "        if (bytes == 0) return false;\n"
     "        std::lock_guard<std::mutex> l(mu);\n"
     "        char *base = nullptr;\n"
     "        if (hipMalloc((void **)&base, bytes) != hipSuccess) {\n"
     "            (void)hipGetLastError();\n"
     "            return false;\n"
     "        }\n"
     "        Slab sl;\n"
     "        sl.base = base;\n"
     "        sl.size = bytes;\n"
     "        sl.free[0] = bytes;\n"
     "        slabs.push_back(std::move(sl));\n"
     "        slab_bytes += (int64_t)bytes;\n"
     "        slab_allocs++;\n"
     "        return true;\n"
     "    }\n"
     "    void trim_locked() { // hand the slabs without a live block back to the device\n"),
    ("    int two_lanes = 1;       // two parts of a batch searched side by side",
     "    int arena_reserve_pct = 0; // LM_ARENA_RESERVE_PCT: share of the scratch budget the arena takes from the device as one slab at the first search (0: slabs on demand)\n"
     "    int two_lanes = 1;       // two parts of a batch searched side by side"),
    ('        if (const char *e = getenv("LM_TWO_LANES")) two_lanes = atoi(e) != 0;\n',
     '        if (const char *e = getenv("LM_TWO_LANES")) two_lanes = atoi(e) != 0;\n'
     '        if (const char *e = getenv("LM_ARENA_RESERVE_PCT")) arena_reserve_pct = std::max(0, std::min(90, atoi(e)));\n'),
    ("    ScratchArena arena;      // phase buffers of the searches on this handle",
     "    bool arena_reserved = false; // LM_ARENA_RESERVE_PCT: the slab was asked for (once per handle, again after the scratch was dropped)\n"
     "    ScratchArena arena;      // phase buffers of the searches on this handle"),
])
edit(os.path.join(csrc, "lm_pipeline.hip"), [
    ("    tls_lane = 0;\n    ix->active_lanes = 1;\n    if (qb->parts.empty()) {\n",
     "    tls_lane = 0;\n    ix->active_lanes = 1;\n"
     "    if (ix->tune.arena_reserve_pct > 0 && !ix->arena_reserved && ix->scratch_budget > 0) { // (see ScratchArena::reserve)\n"
     "        ix->arena_reserved = true;\n"
     "        const bool ok = ix->arena.reserve((size_t)(ix->scratch_budget / 100 * ix->tune.arena_reserve_pct));\n"
     "        if (getenv(\"LM_DEBUG\")) fprintf(stderr, \"[lm] scratch arena: %d %% of the budget reserved as one slab: %s\\n\", ix->tune.arena_reserve_pct, ok ? \"yes\" : \"refused\");\n"
     "    }\n"
     "    if (qb->parts.empty()) {\n"),
    ("    ix->arena.trim();\n    if (getenv(\"LM_DEBUG\")) fprintf(stderr, \"[lm] device scratch of lane %d dropped after: %s\\n\", tls_lane, why);\n",
     "    ix->arena.trim();\n    ix->arena_reserved = false; // (an empty reserved slab went back with the trim: the next search asks again)\n"
     "    if (getenv(\"LM_DEBUG\")) fprintf(stderr, \"[lm] device scratch of lane %d dropped after: %s\\n\", tls_lane, why);\n"),
])
print("LM_ARENA_RESERVE_PCT adopted under", root)

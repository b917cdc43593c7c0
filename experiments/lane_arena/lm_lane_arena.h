// experiments/lane_arena/lm_lane_arena.h - STAGED for round 5, not in lexicmap_amd/csrc: the scratch of a two-lane handle as
// fixed slabs cut ONCE from the scratch budget.
//
// Why (profiles/r04_c3_steady.json): two lanes are worth 10 % at C3, but both lanes carve their phase
// buffers out of ONE arena that grows by hipMalloc on demand.  A fresh handle needs three C3 steps to settle (17.5, 14.5, then
// 12.2 s): hipMalloc / hipFree synchronise the device, so one lane's allocation waits for the other lane's persistent WFA
// kernels; while the slabs of both lanes' peaks pile up the device runs out, allocations fail, batch parts are halved and stay
// halved; and the serialised measurement step (one lane with the whole budget) re-cuts everything and halves further.
//
// Here the handle owns two device slabs of half the arena budget each, allocated once.  With two lanes every lane's arena
// works inside its own slab (the lanes' allocation sequences no longer interact: each settles like the single-lane case did);
// with one lane the arena of lane 0 works inside both (the largest phase buffer is 26 % of the budget, so no single buffer
// needs more than one slab).  A request no slab can take goes to an overflow slab from the device (today's behaviour, counted),
// which trim() hands back.  Same interface as lm::ScratchArena (alloc / release / trim), so DBuf needs no change.
//
// Integration (round 5): lm_index gets `LaneSlabs slabs; LaneArena arena[2]`; every `tls_arena = &ix->arena` becomes
// `&ix->arena[tls_lane]`; search_parts calls `ix->slabs.assign(ix->arena[0], ix->arena[1], lanes)` after `active_lanes` is
// set (all phase buffers are released between searches: release_big); lm_set_scratch_budget reserves the slabs (90 % of the
// scratch budget: small and non-phase buffers stay plain allocations); drop_scratch / close drop them.
#pragma once
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

namespace lm {

struct LaneArena {
    struct Slab {
        char *base = nullptr;
        size_t size = 0;
        bool fixed = false;            // one of the handle's slabs: never freed by trim()
        std::map<size_t, size_t> free; // offset -> length
    };
    std::mutex mu;
    std::vector<Slab> slabs;
    std::unordered_map<void *, std::pair<int, size_t>> live; // block -> (slab, length)
    int64_t live_bytes = 0, overflow_allocs = 0, overflow_bytes = 0;
    static constexpr size_t ALIGN = 4096;
    ~LaneArena() { trim(); }

    void adopt(char *base, size_t size) { // a fixed slab of the handle, whole and free
        std::lock_guard<std::mutex> l(mu);
        Slab s;
        s.base = base;
        s.size = size;
        s.fixed = true;
        s.free[0] = size;
        put(std::move(s));
    }
    // gives the fixed slabs back to the handle; every block must have been released (between searches they are)
    void drop_fixed() {
        std::lock_guard<std::mutex> l(mu);
        for (auto &s : slabs)
            if (s.base && s.fixed) {
                if (!(s.free.size() == 1 && s.free.begin()->second == s.size)) throw std::runtime_error("LaneArena::drop_fixed: a block is still live");
                s = Slab();
            }
    }
    void *alloc(size_t bytes) { // throws DeviceOOM
        bytes = (bytes + ALIGN - 1) / ALIGN * ALIGN;
        std::lock_guard<std::mutex> l(mu);
        for (int pass = 0; pass < 2; pass++) {
            int bs = -1;
            size_t boff = 0, blen = ~(size_t)0;
            for (size_t si = 0; si < slabs.size(); si++)
                for (auto &f : slabs[si].free)
                    if (f.second >= bytes && f.second < blen) { // best fit over all slabs
                        bs = (int)si;
                        boff = f.first;
                        blen = f.second;
                    }
            if (bs >= 0) {
                Slab &sl = slabs[bs];
                sl.free.erase(boff);
                if (blen > bytes) sl.free[boff + bytes] = blen - bytes;
                void *p = sl.base + boff;
                live[p] = {bs, bytes};
                live_bytes += (int64_t)bytes;
                return p;
            }
            if (pass == 1) break;
            // overflow: a slab of its own from the device (what every allocation was before); empty ones first
            char *base = nullptr;
            hipError_t e = hipMalloc((void **)&base, bytes);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                trim_locked();
                e = hipMalloc((void **)&base, bytes);
            }
            if (e != hipSuccess) {
                (void)hipGetLastError();
                throw DeviceOOM("device scratch allocation of " + std::to_string(bytes >> 20) + " MB failed: the lane's slabs are full and the device has no room for an overflow slab");
            }
            Slab s;
            s.base = base;
            s.size = bytes;
            s.free[0] = bytes;
            put(std::move(s));
            overflow_allocs++;
            overflow_bytes += (int64_t)bytes;
        }
        throw DeviceOOM("scratch arena: internal error");
    }
    bool release(void *p) {
        std::lock_guard<std::mutex> l(mu);
        auto it = live.find(p);
        if (it == live.end()) return false;
        Slab &sl = slabs[it->second.first];
        size_t off = (size_t)((char *)p - sl.base), len = it->second.second;
        live_bytes -= (int64_t)len;
        live.erase(it);
        auto nx = sl.free.lower_bound(off);
        if (nx != sl.free.end() && off + len == nx->first) {
            len += nx->second;
            nx = sl.free.erase(nx);
        }
        if (nx != sl.free.begin()) {
            auto pv = std::prev(nx);
            if (pv->first + pv->second == off) {
                pv->second += len;
                return true;
            }
        }
        sl.free[off] = len;
        return true;
    }
    void trim() {
        std::lock_guard<std::mutex> l(mu);
        trim_locked();
    }

  private:
    void put(Slab &&s) {
        for (auto &x : slabs)
            if (!x.base) {
                x = std::move(s);
                return;
            }
        slabs.push_back(std::move(s));
    }
    void trim_locked() { // the empty OVERFLOW slabs go back to the device; the handle's slabs stay
        for (auto &sl : slabs)
            if (sl.base && !sl.fixed && sl.free.size() == 1 && sl.free.begin()->second == sl.size) {
                (void)hipFree(sl.base);
                overflow_bytes -= (int64_t)sl.size;
                sl = Slab();
            }
    }
};

// the handle's two slabs: allocated once, handed to the lane arenas according to the number of lanes of a search
struct LaneSlabs {
    char *base[2] = {nullptr, nullptr};
    size_t size[2] = {0, 0};
    int assigned_lanes = 0; // 0: with nobody
    // two slabs of bytes / 2 each; returns false (and holds nothing) when the device refuses: the arenas then work from
    // overflow slabs only, i.e. exactly as before
    bool reserve(size_t bytes) {
        drop();
        const size_t half = bytes / 2 / LaneArena::ALIGN * LaneArena::ALIGN;
        if (half == 0) return false;
        for (int i = 0; i < 2; i++)
            if (hipMalloc((void **)&base[i], half) != hipSuccess) {
                (void)hipGetLastError();
                base[i] = nullptr;
                drop();
                return false;
            } else {
                size[i] = half;
            }
        return true;
    }
    // between searches (no live block): lanes == 2 -> one slab each; lanes == 1 -> both to a0
    void assign(LaneArena &a0, LaneArena &a1, int lanes) {
        if (!base[0] || lanes == assigned_lanes) return;
        if (assigned_lanes) {
            a0.drop_fixed();
            a1.drop_fixed();
        }
        a0.adopt(base[0], size[0]);
        (lanes == 2 ? a1 : a0).adopt(base[1], size[1]);
        assigned_lanes = lanes;
    }
    void unassign(LaneArena &a0, LaneArena &a1) {
        if (assigned_lanes) {
            a0.drop_fixed();
            a1.drop_fixed();
        }
        assigned_lanes = 0;
    }
    void drop() { // (after unassign)
        for (int i = 0; i < 2; i++) {
            if (base[i]) (void)hipFree(base[i]);
            base[i] = nullptr;
            size[i] = 0;
        }
        assigned_lanes = 0;
    }
};

} // namespace lm

// tests/format_host.cpp - the product's host-side chunk decoder (lexicmap_amd/csrc/lm_format.cpp: decode_seed_chunk, what the
// loader's threads run) behind a C interface for the CPU test: decodes one seeds/chunk_NNN.bin into its flat (k-mer, value,
// mask) arrays, re-using ONE SeedChunk across calls exactly as the loader's slots are re-used.
#include <cstring>
#include <string>

#include "../lexicmap_amd/csrc/lm_format.h"

static lm::HostIndex g_idx;
static lm::SeedChunk g_chunk;
static std::string g_err;

extern "C" {
const char *fh_error() { return g_err.c_str(); }
int fh_open(const char *dir, int shard_rank, int shard_count) {
    int st = 0;
    g_idx = lm::HostIndex();
    g_err = lm::load_index(dir, shard_rank, shard_count, g_idx, st);
    return g_err.empty() ? 0 : st;
}
int fh_nfiles() { return (int)g_idx.seed_files.size(); }
const char *fh_file(int i) { return g_idx.seed_files[(size_t)i].c_str(); }
// returns the number of seeds (>= 0) or -status; the arrays stay valid until the next call
long long fh_decode(const char *path, const unsigned long long **kmers, const unsigned long long **vals, const unsigned short **masks) {
    int st = 0, ap = -1;
    g_err = lm::decode_seed_chunk(path, g_idx, g_chunk, st, ap);
    if (!g_err.empty()) return -(long long)(st ? st : 1);
    *kmers = (const unsigned long long *)g_chunk.kmers.data();
    *vals = (const unsigned long long *)g_chunk.vals.data();
    *masks = g_chunk.masks.data();
    return (long long)g_chunk.n;
}
}

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
// the same with the genome bases handed to a SINK (what the loader does: to the device) instead of appended to the host store;
// the sink here writes them where they belong in g_sunk and counts its calls
static std::vector<unsigned char> g_sunk;
static long g_sink_calls = 0;
int fh_open_sink(const char *dir, int shard_rank, int shard_count) {
    int st = 0;
    g_idx = lm::HostIndex();
    g_sunk.clear();
    g_sink_calls = 0;
    g_err = lm::load_index(dir, shard_rank, shard_count, g_idx, st, false);
    if (!g_err.empty()) return st;
    g_sunk.assign(g_idx.gbits_bound + 64, 0); // (zero-filled, as the loader's device buffer is: the padding behind a genome)
    g_idx.gbits_sink = [](const uint8_t *src, size_t n, int64_t off) {
        if (off < 0 || (size_t)off + n > g_sunk.size()) return false;
        memcpy(g_sunk.data() + off, src, n);
        g_sink_calls++;
        return true;
    };
    g_err = lm::load_index_genomes(dir, g_idx, st);
    g_idx.gbits_sink = nullptr;
    if (g_err.empty()) g_idx.gbits.assign(g_sunk.begin(), g_sunk.begin() + g_idx.gbits_total); // (fh_genome reads g_idx.gbits)
    return g_err.empty() ? 0 : st;
}
long fh_sink_calls() { return g_sink_calls; }
long long fh_store_bytes() { return (long long)g_idx.gbits.size(); }
const unsigned char *fh_store() { return g_idx.gbits.data(); }
int fh_nfiles() { return (int)g_idx.seed_files.size(); }
// the genome store load_index_genomes left: local genomes (bases on this shard) and the others (names and contig tables only)
int fh_ngenomes() { return (int)g_idx.genomes.size(); }
int fh_nothers() { return (int)g_idx.others.size(); }
// local genome i: its key, length, contigs and the address of its packed bases (2 bits per base, first base in bits 7-6)
const unsigned char *fh_genome(int i, unsigned long long *bg, int *len, int *nseqs, const char **id) {
    const lm::HostGenome &g = g_idx.genomes[(size_t)i];
    *bg = g.bg;
    *len = g.len;
    *nseqs = g.nseqs;
    *id = g.id.c_str();
    return g_idx.gbits.data() + g.bits_off;
}
void fh_other(int i, unsigned long long *bg, int *len, int *nseqs, const char **id) {
    const lm::HostGenome &g = g_idx.others[(size_t)i];
    *bg = g.bg;
    *len = g.len;
    *nseqs = g.nseqs;
    *id = g.id.c_str();
}
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

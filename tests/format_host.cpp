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

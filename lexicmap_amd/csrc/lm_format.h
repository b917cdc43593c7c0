// lm_format.h — host-side readers of the reference on-disk index (format 3.x) used to build the HBM image.
// Reference: info.toml (lib-index-build.go:1914-1932), seeds/chunk_NNN.bin(.idx) (kv/kv-data.go:66-125,394-562;
// kv/kv-reader.go:762-1021), genomes/batch_NNNN/genomes.bin(.idx) (genome/genome.go:184-358,388-474),
// genomes.map.bin (lib-index-build.go:649-655,1969-2016).  masks.bin: this build's own layout or a headered big-endian mask
// list (the upstream lexichash layout is not in the reference tree; see lm_format.cpp and DESIGN.md).
#pragma once
#include <stdint.h>
#include <functional>
#include <string>
#include <unordered_map>
#include <vector>

namespace lm {

struct HostGenome {
    uint64_t bg = 0;        // batch<<17 | index in batch
    int64_t global = 0;     // dense number over the whole index
    std::string id;         // from genomes.map.bin
    int32_t genome_size = 0, len = 0, nseqs = 0;
    std::vector<int32_t> seq_sizes;
    std::vector<std::string> seq_ids;
    int64_t bits_off = 0;   // byte offset of the packed sequence in HostIndex::gbits
};

struct HostIndex {
    int k = 0, M = 0, mask_prefix = 0, anchor_prefix = 0;
    int main_version = 0, minor_version = 0;
    int64_t total_bases = 0;
    int contig_interval = 1000;
    int genome_batches = 0;
    std::vector<uint64_t> masks;
    std::vector<std::string> seed_files; // seeds/chunk_NNN.bin, sorted
    // genomes of this shard
    std::vector<HostGenome> genomes;
    // names / contig tables of the genomes held by the OTHER shards (no bases), keyed by batch<<17|index
    std::vector<HostGenome> others;
    std::unordered_map<uint64_t, int> other_of;
    // genome chunks (genomes.chunks.bin, lib-index-search.go:504-538): key -> {list number, #chunks, chunk index}
    struct ChunkInfo {
        int list, n, idx;
    };
    std::unordered_map<uint64_t, ChunkInfo> chunk_of;
    bool has_chunks = false;
    // dense genome number -> local number on this shard (-1: another shard's); empty when unsharded. The chunks of a split
    // genome stay together: a genome goes to shard (dense number of its FIRST chunk) % shard_count.
    std::vector<int32_t> g2local;
    bool synthetic = false;        // built by lm_index_build_synthetic: names are a function of the genome number
    int64_t synth_genomes = 0;
    int32_t synth_genome_len = 0;
    std::vector<uint8_t> gbits;    // 2-bit packed, first base in bits 7-6; each genome padded to 8 bytes
    std::vector<int64_t> batch_first; // [batches+1] global dense number of the first genome of each batch
    int shard_rank = 0, shard_count = 1;
    int64_t n_local_genomes = 0, max_genome_len = 1; // from the batch .idx files (before the batches themselves are read)
    size_t gbits_bound = 0;        // upper bound of the packed bytes (+ padding) of this shard's genomes
    // Optional: where load_index_genomes puts the packed bases of a local genome instead of appending them to `gbits` (the
    // loader: straight from the read buffer to their place on the device).  A sink that returns false stops the load (status 2).
    std::function<uint8_t *(size_t bytes)> gbits_buffer; // the buffer a run of records is read into (the loader: pinned memory)
    std::function<bool(const uint8_t *src, size_t nbytes, int64_t bits_off)> gbits_sink;
    std::function<void()> gbits_batch_end;               // the buffer is about to be read into again
    int64_t gbits_total = 0;       // bytes of the store once every batch is read (= gbits.size() without a sink)
};

// Everything of the index except the seeds (info.toml, masks, genomes, id map, chunk lists) + the list of seed chunk files.
// returns empty string on success, else the error text. status: 1 io, 2 format
std::string load_index(const std::string &dir, int shard_rank, int shard_count, HostIndex &out, int &status, bool genomes_now = true);
// genomes_now = false leaves the genome batches (names, contig tables, bases) to this second call, which may run on another
// thread beside the seed passes: n_local_genomes / max_genome_len / batch_first / g2local are already set by load_index
std::string load_index_genomes(const std::string &dir, HostIndex &out, int &status);

// One seeds/chunk_NNN.bin decoded into flat arrays (k-mer, value, mask) of the seeds whose genome is on this shard: what the
// seed packer is shown.  The reference reads these files with one goroutine per file (kv-reader.go:762-1021); the loader
// streams them - decode, hand to the packer, drop - so the host never holds more than a few files.
// A SeedChunk is REUSED for file after file: its arrays only grow (`n` seeds are valid), so that after the first files no
// decode touches fresh pages - zero-filling and faulting in 18 B per seed of new memory per file cost more than the decoding.
struct SeedChunk {
    std::vector<uint64_t> kmers, vals;
    std::vector<uint16_t> masks;
    std::vector<uint8_t> file; // the file's bytes (+ 16 of padding)
    size_t n = 0;
    // set by the owner before the first decode: the arrays are cut for at least this many seeds / file bytes at once (the
    // loader: what its LARGEST file needs), so that they never move afterwards - they are registered with the driver
    size_t min_seeds = 0, min_file_bytes = 0;
};
std::string decode_seed_chunk(const std::string &path, const HostIndex &idx, SeedChunk &out, int &status, int &anchor_prefix);

} // namespace lm

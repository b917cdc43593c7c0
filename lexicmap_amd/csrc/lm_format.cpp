// lm_format.cpp — see lm_format.h
#include "lm_format.h"

#include <chrono>
#include <algorithm>
#include <atomic>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <dirent.h>
#include <fcntl.h>
#include <map>
#include <sys/stat.h>
#include <unistd.h>

namespace lm {

namespace {

struct File {
    FILE *f = nullptr;
    explicit File(const std::string &p) { f = fopen(p.c_str(), "rb"); }
    ~File() {
        if (f) fclose(f);
    }
    bool ok() const { return f != nullptr; }
    bool read(void *dst, size_t n) { return fread(dst, 1, n, f) == n; }
};

inline uint64_t be64(const uint8_t *b) {
    uint64_t v = 0;
    for (int i = 0; i < 8; i++) v = (v << 8) | b[i];
    return v;
}
inline uint32_t be32(const uint8_t *b) { return ((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | b[3]; }
inline uint32_t be16(const uint8_t *b) { return ((uint32_t)b[0] << 8) | b[1]; }

bool read_all(const std::string &path, std::vector<uint8_t> &buf) {
    File f(path);
    if (!f.ok()) return false;
    fseek(f.f, 0, SEEK_END);
    long n = ftell(f.f);
    fseek(f.f, 0, SEEK_SET);
    buf.resize((size_t)n);
    return n == 0 || f.read(buf.data(), (size_t)n);
}
struct FdGuard {
    int fd;
    explicit FdGuard(int f) : fd(f) {}
    ~FdGuard() {
        if (fd >= 0) close(fd);
    }
};
bool pread_all(int fd, uint8_t *dst, size_t n, size_t off) {
    while (n) {
        const ssize_t r = pread(fd, dst, n, (off_t)off);
        if (r <= 0) return false;
        dst += r;
        off += (size_t)r;
        n -= (size_t)r;
    }
    return true;
}
// n bytes at `off` by up to 8 threads (one thread copies out of the page cache at 2-3 GB/s)
bool pread_all_mt(int fd, uint8_t *dst, size_t n, size_t off) {
    const size_t piece = (size_t)32 << 20;
    const size_t npieces = (n + piece - 1) / piece;
    if (npieces <= 1) return pread_all(fd, dst, n, off);
    const unsigned nt = (unsigned)std::min<size_t>(npieces, std::max(1u, std::min(8u, std::thread::hardware_concurrency())));
    std::atomic<size_t> next{0};
    std::atomic<bool> ok{true};
    auto body = [&]() {
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= npieces) break;
            const size_t o = i * piece, len = std::min(piece, n - o);
            if (!pread_all(fd, dst + o, len, off + o)) ok = false;
        }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; t++) th.emplace_back(body);
    body();
    for (auto &t : th) t.join();
    return ok;
}

long long toml_int(const std::string &text, const char *key, long long dflt) {
    size_t kl = strlen(key), pos = 0;
    while (pos < text.size()) {
        size_t eol = text.find('\n', pos);
        if (eol == std::string::npos) eol = text.size();
        if (text.compare(pos, kl, key) == 0) {
            size_t q = pos + kl;
            while (q < eol && text[q] == ' ') q++;
            if (q < eol && text[q] == '=') return atoll(text.c_str() + q + 1);
        }
        pos = eol + 1;
    }
    return dflt;
}

// group varint of two uint64 (util/varint-GB.go:88-113: control byte = the two byte lengths - 1), read from a buffer with >= 16
// readable bytes behind every record (decode_seed_chunk pads its copy of the file): two unaligned 8-byte big-endian loads
inline uint64_t load_be64(const uint8_t *p) {
    uint64_t v;
    memcpy(&v, p, 8);
    return __builtin_bswap64(v);
}
inline int gv2_fast(uint8_t ctrl, const uint8_t *p, uint64_t &a, uint64_t &b) {
    const int l1 = ((ctrl >> 3) & 7) + 1, l2 = (ctrl & 7) + 1;
    a = load_be64(p) >> (64 - 8 * l1);
    b = load_be64(p + l1) >> (64 - 8 * l2);
    return l1 + l2;
}

} // namespace

// Decodes one seeds chunk (kv-data.go:66-89 layout) into flat (k-mer, value, mask) arrays, keeping the seeds whose genome
// is on this shard.
std::string decode_seed_chunk(const std::string &path, const HostIndex &idx, SeedChunk &out, int &status,
                              int &anchor_prefix_out) {
    const std::vector<int64_t> &batch_first = idx.batch_first;
    out.n = 0;
    std::vector<uint8_t> &buf = out.file;
    size_t file_bytes = 0;
    {   // the file's bytes + 16 of padding (the 8-byte loads of gv2_fast / the value reads may look past the last record)
        File f(path);
        if (!f.ok()) {
            status = 1;
            return "cannot read " + path;
        }
        fseek(f.f, 0, SEEK_END);
        const long nb = ftell(f.f);
        fseek(f.f, 0, SEEK_SET);
        if (nb < 0) {
            status = 1;
            return "cannot read " + path;
        }
        file_bytes = (size_t)nb;
        if (buf.size() < file_bytes + 16) buf.resize(std::max(file_bytes, out.min_file_bytes) + 16);
        if (file_bytes && !f.read(buf.data(), file_bytes)) {
            status = 1;
            return "cannot read " + path;
        }
        memset(buf.data() + file_bytes, 0, 16);
    }
    if (file_bytes < 32 || memcmp(buf.data(), ".kv-data", 8) != 0) {
        status = 2;
        return "k-mer-value data: invalid binary format: " + path;
    }
    if (buf[8] != 1) {
        status = 2;
        return "k-mer-value data: version mismatch: " + path;
    }
    const bool use7 = (buf[11] & 1) != 0;
    const int nvb = use7 ? 7 : 8;
    int64_t mask0 = (int64_t)be64(&buf[16]), nmask = (int64_t)be64(&buf[24]);
    // anchor prefix from the .idx header (the data itself does not need the index)
    {
        File fi(path + ".idx");
        uint8_t h[32];
        if (!fi.ok() || !fi.read(h, 32) || memcmp(h, ".kvindex", 8) != 0) {
            status = 2;
            return "k-mer-value index: invalid binary format: " + path + ".idx";
        }
        if ((int)h[11] != idx.mask_prefix) {
            status = 2;
            return "lengths of mask prefix mismatch between info.toml and the seed data";
        }
        anchor_prefix_out = h[12]; // users might have run 'utils reindex-seeds' (lib-index-search.go:611)
    }
    size_t p = 32;
    const size_t n = file_bytes;
    const int sc = idx.shard_count;
    // every seed is at least nvb bytes of the file: the arrays are cut once (and kept: SeedChunk)
    const size_t ub = (n - 32) / (size_t)nvb + 1;
    if (out.kmers.size() < ub) {
        const size_t cut = std::max(ub, out.min_seeds);
        out.kmers.resize(cut);
        out.vals.resize(cut);
        out.masks.resize(cut);
    }
    uint64_t *ok = out.kmers.data(), *ov = out.vals.data();
    uint16_t *om = out.masks.data();
    size_t ns = 0;
    for (int64_t im = 0; im < nmask; im++) {
        if (p + 8 > n) {
            status = 2;
            return "k-mer-value data: broken file: " + path;
        }
        uint64_t nk = be64(&buf[p]);
        p += 8;
        if (nk == 0) continue;
        if (mask0 + im >= idx.M) {
            status = 2;
            return "k-mer-value data: mask number out of range: " + path;
        }
        const uint16_t mk = (uint16_t)(mask0 + im);
        uint64_t off = 0;
        for (;;) {
            // a record = ctrl byte + two group-varint values (<= 16 bytes), twice: k-mer deltas, then value counts
            if (p + 1 > n || p + 1 + (size_t)(((buf[p] >> 3) & 7) + (buf[p] & 7) + 2) > n) {
                status = 2;
                return "k-mer-value data: broken file: " + path;
            }
            uint8_t ctrl = buf[p++];
            bool last_pair = (ctrl & 128) != 0, has2 = (ctrl & 64) == 0;
            ctrl &= 63;
            uint64_t d1, d2, l1, l2;
            p += gv2_fast(ctrl, &buf[p], d1, d2);
            uint64_t k1 = d1 + off, k2 = k1 + d2;
            off = k2;
            if (p + 1 > n || p + 1 + (size_t)(((buf[p] >> 3) & 7) + (buf[p] & 7) + 2) > n) {
                status = 2;
                return "k-mer-value data: broken file: " + path;
            }
            ctrl = buf[p++];
            p += gv2_fast(ctrl, &buf[p], l1, l2);
            for (int w = 0; w < 2; w++) {
                if (w == 1 && last_pair && !has2) break;
                uint64_t kmer = w == 0 ? k1 : k2, lv = w == 0 ? l1 : l2;
                if (lv > n || p + lv * nvb > n) {
                    status = 2;
                    return "k-mer-value data: broken file: " + path;
                }
                for (uint64_t j = 0; j < lv; j++) {
                    uint64_t v = use7 ? load_be64(&buf[p]) >> 8 : load_be64(&buf[p]);
                    p += nvb;
                    if (sc > 1) {
                        uint64_t bg = v >> 30;
                        uint64_t batch = bg >> 17, gi = bg & 0x1ffff;
                        int64_t g = (batch + 1 < batch_first.size() ? batch_first[batch] : 0) + (int64_t)gi;
                        if (g < 0 || g >= (int64_t)idx.g2local.size() || idx.g2local[(size_t)g] < 0) continue;
                    }
                    ok[ns] = kmer;
                    ov[ns] = v;
                    om[ns] = mk;
                    ns++;
                }
            }
            if (last_pair) break;
        }
    }
    out.n = ns;
    return "";
}

std::string load_index(const std::string &dir, int shard_rank, int shard_count, HostIndex &out, int &status, bool genomes_now) {
    status = 0;
    if (shard_count < 1) shard_count = 1;
    out.shard_rank = shard_rank;
    out.shard_count = shard_count;
    std::vector<uint8_t> buf;
    if (!read_all(dir + "/info.toml", buf)) {
        status = 1;
        return "failed to read index info file: " + dir + "/info.toml";
    }
    std::string text((const char *)buf.data(), buf.size());
    out.main_version = (int)toml_int(text, "main-version", -1);
    out.minor_version = (int)toml_int(text, "minor-version", 0);
    if (out.main_version != 3) { // lib-index-search.go:290-292
        status = 2;
        return "index main versions do not match";
    }
    if (out.minor_version < 5) {
        // lib-index-search.go:1212-1215: an index older than format 3.5 is searched with lexichash's
        // MaskKnownDistinctPrefixesWithStrandBias, which lives in the un-vendored lexichash module and is not restated here:
        // searching it with the 3.5 masking would silently return other seeds than the reference.  Refused, not guessed.
        status = 2;
        return "index format 3." + std::to_string(out.minor_version) + " (minor-version < 5) needs the strand-biased masking of "
               "lib-index-search.go:1212-1215 (MaskKnownDistinctPrefixesWithStrandBias), which this build does not implement: "
               "re-create the index with lexicmap >= v0.5.0 (" + dir + ")";
    }
    out.total_bases = toml_int(text, "input-bases", 0);
    out.contig_interval = (int)toml_int(text, "contig-interval", 1000);
    out.genome_batches = (int)toml_int(text, "genome-batches", 1);
    int partitions = (int)toml_int(text, "index-partitions", 4096);

    // masks.bin.  Two layouts are accepted:
    //  (a) this build's own (magic LMMASKS1: k u8 at 8, count u32 at 12, masks from 24), written by the oracle's index writer;
    //  (b) a headered list of big-endian 64-bit masks as lexichash's WriteToFile is believed to produce (lexichash v0.5.x is
    //      not in the reference tree, so this cannot be verified against a real file): 8-byte magic, 8 meta bytes with k at
    //      byte 10, u64 count, u64 seed, then the masks.  It is only accepted when k and the count agree with info.toml
    //      (max-K, masks), the size is exact and the masks are strictly ascending, else the open fails with LM_ERR_FORMAT.
    if (!read_all(dir + "/masks.bin", buf) || buf.size() < 24) {
        status = buf.empty() ? 1 : 2;
        return "failed to read masks: " + dir + "/masks.bin";
    }
    size_t mask_at = 24;
    if (memcmp(buf.data(), "LMMASKS1", 8) == 0) {
        out.k = buf[8];
        out.M = (int)be32(&buf[12]);
    } else {
        const long long tk = toml_int(text, "max-K", -1), tm = toml_int(text, "masks", -1);
        bool ok = false;
        if (buf.size() >= 32 && tk >= 1 && tk <= 32 && tm >= 1 && buf.size() == 32 + (size_t)tm * 8 && (long long)buf[10] == tk &&
            (long long)be64(&buf[16]) == tm) {
            ok = true;
            for (long long i = 1; i < tm && ok; i++) ok = be64(&buf[32 + (size_t)i * 8]) > be64(&buf[32 + (size_t)(i - 1) * 8]);
        }
        if (!ok) {
            status = 2;
            return "masks.bin: unknown layout (neither this build's LMMASKS1 nor a k/count-consistent lexichash mask list): " + dir;
        }
        out.k = (int)tk;
        out.M = (int)tm;
        mask_at = 32;
        // not silently: this layout was inferred, no index written by the reference's own `lexicmap index` has been seen
        fprintf(stderr, "[lexicmap_hip] warning: %s/masks.bin is not in this build's LMMASKS1 layout; read as a headered big-endian "
                        "mask list (k = %d, %d masks, ascending) - the lexichash file layout is not verified against an upstream "
                        "file (DESIGN.md section 5)\n", dir.c_str(), out.k, out.M);
    }
    if (buf.size() < mask_at + (size_t)out.M * 8 || out.k < 1 || out.k > 32) {
        status = 2;
        return "broken masks file";
    }
    out.masks.resize(out.M);
    for (int i = 0; i < out.M; i++) out.masks[i] = be64(&buf[mask_at + (size_t)i * 8]);
    {   // lib-index-search.go:467-469
        int p = (int)(std::log2((double)out.M) / 2);
        out.mask_prefix = p < 1 ? 1 : p;
        int a = (int)(std::log2((double)partitions) / 2);
        out.anchor_prefix = a < 1 ? 1 : a;
    }

    // genomes: batch_NNNN/genomes.bin + .idx
    out.batch_first.assign(out.genome_batches + 1, 0);
    // pass 1: genomes per batch -> dense numbers; genome chunk lists -> shard of every genome
    size_t gbits_bound = 0; // packed bytes of all genomes (+ padding): the store is cut once, not grown batch by batch
    std::vector<int32_t> rec_len; // bases of every genome record, dense order (from the .idx files)
    {
        int64_t g0 = 0;
        for (int b = 0; b < out.genome_batches; b++) {
            char name[64];
            snprintf(name, sizeof name, "/genomes/batch_%04d/genomes.bin.idx", b);
            std::vector<uint8_t> ib;
            if (!read_all(dir + name, ib) || ib.size() < 24 || memcmp(ib.data(), ".genomei", 8) != 0) {
                status = ib.empty() ? 1 : 2;
                return std::string("genome data: invalid binary format: ") + name;
            }
            out.batch_first[b] = g0;
            const uint32_t nrec = be32(&ib[20]);
            g0 += nrec;
            for (uint32_t r = 0; r < nrec && 24 + (size_t)r * 12 + 12 <= ib.size(); r++) { // bases per record -> packed bytes, padded
                const uint32_t bases = be32(&ib[24 + (size_t)r * 12 + 8]);
                gbits_bound += ((size_t)bases + 3) / 4 + 16;
                rec_len.push_back((int32_t)bases);
            }
        }
        out.batch_first[out.genome_batches] = g0;
        std::vector<int64_t> canon((size_t)g0);
        for (int64_t g = 0; g < g0; g++) canon[(size_t)g] = g;
        std::vector<uint8_t> cb;
        if (read_all(dir + "/genomes.chunks.bin", cb) && !cb.empty()) { // readGenomeChunksLists, lib-index-build.go:2193-2245
            size_t p = 0;
            int list = 0;
            while (p + 8 <= cb.size()) {
                const uint64_t n = be64(&cb[p]);
                p += 8;
                if (n > (cb.size() - p) / 8) {
                    status = 2;
                    return "broken genome chunk file";
                }
                int64_t first = -1;
                for (uint64_t j = 0; j < n; j++, p += 8) {
                    const uint64_t key = be64(&cb[p]);
                    out.chunk_of[key] = HostIndex::ChunkInfo{list, (int)n, (int)j};
                    const uint64_t batch = key >> 17, gi = key & 0x1ffff;
                    if (batch >= (uint64_t)out.genome_batches) continue;
                    const int64_t g = out.batch_first[batch] + (int64_t)gi;
                    if (g >= g0) continue;
                    if (first < 0) first = g;
                    canon[(size_t)g] = first;
                }
                list++;
            }
            out.has_chunks = !out.chunk_of.empty();
        }
        if (shard_count > 1) {
            out.g2local.assign((size_t)g0, -1);
            int32_t nl = 0;
            for (int64_t g = 0; g < g0; g++)
                if ((int)(canon[(size_t)g] % shard_count) == shard_rank) out.g2local[(size_t)g] = nl++;
        }
        // what the seed packer needs of the genomes, known before their batch files are read (load_index_genomes)
        out.n_local_genomes = 0;
        out.max_genome_len = 1;
        for (int64_t g = 0; g < g0 && (size_t)g < rec_len.size(); g++)
            if (out.g2local.empty() || out.g2local[(size_t)g] >= 0) {
                out.n_local_genomes++;
                out.max_genome_len = std::max<int64_t>(out.max_genome_len, rec_len[(size_t)g]);
            }
    }
    if (shard_count > 1) { // exactly this shard's genomes (their lengths are in the batch indexes)
        gbits_bound = 64;
        for (size_t g = 0; g < rec_len.size(); g++)
            if (out.g2local.empty() || ((size_t)g < out.g2local.size() && out.g2local[g] >= 0)) gbits_bound += ((size_t)rec_len[g] + 3) / 4 + 16;
    }
    out.gbits_bound = gbits_bound;
    // seeds
    std::vector<std::string> files;
    {
        DIR *d = opendir((dir + "/seeds").c_str());
        if (!d) {
            status = 1;
            return "seeds file not found in: " + dir + "/seeds";
        }
        while (dirent *de = readdir(d)) {
            std::string nm = de->d_name;
            if (nm.size() > 4 && nm.compare(nm.size() - 4, 4, ".bin") == 0) files.push_back(nm);
        }
        closedir(d);
        std::sort(files.begin(), files.end());
    }
    if (files.empty()) {
        status = 1;
        return "seeds file not found in: " + dir + "/seeds";
    }
    out.seed_files.clear();
    for (auto &f : files) out.seed_files.push_back(dir + "/seeds/" + f);
    {   // the anchor prefix of the seed data (users might have run 'utils reindex-seeds', lib-index-search.go:611)
        File fi(out.seed_files[0] + ".idx");
        uint8_t h[32];
        if (!fi.ok() || !fi.read(h, 32) || memcmp(h, ".kvindex", 8) != 0) {
            status = 2;
            return "k-mer-value index: invalid binary format: " + out.seed_files[0] + ".idx";
        }
        if ((int)h[11] != out.mask_prefix) {
            status = 2;
            return "lengths of mask prefix mismatch between info.toml and the seed data";
        }
        out.anchor_prefix = h[12];
    }
    if (genomes_now) return load_index_genomes(dir, out, status);
    return "";
}

// The genome batches (genomes/batch_NNNN/genomes.bin: names, contig tables, 2-bit bases) and the id map: everything of the
// index that the seed packer does not need - the loader reads them on a host thread while the seed chunks are decoded and
// packed (they were a third of the time an open took).
std::string load_index_genomes(const std::string &dir, HostIndex &out, int &status) {
    std::vector<uint8_t> buf;
    std::map<uint64_t, std::string> id_of;
    if (read_all(dir + "/genomes.map.bin", buf)) {
        size_t p = 0;
        while (p + 2 <= buf.size()) {
            uint32_t l = be16(&buf[p]);
            p += 2;
            if (p + l + 8 > buf.size()) break;
            std::string id((const char *)&buf[p], l);
            p += l;
            id_of[be64(&buf[p])] = id;
            p += 8;
        }
    } else {
        status = 1;
        return "failed to read " + dir + "/genomes.map.bin";
    }
    int64_t global = 0, run = 0; // run: bytes of the store so far when a sink takes the bases
    // (LM_DEBUG: where the reader's time goes - the reads into its buffer, the records handed to the sink, the waits for the sink)
    const bool rdbg = getenv("LM_DEBUG") != nullptr;
    double t_read = 0, t_sink = 0, t_end = 0, t_idx = 0;
    auto nowms = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const bool to_sink = (bool)out.gbits_sink;
    if (!to_sink) out.gbits.reserve(out.gbits_bound);
    // A batch file (GBs) is read in RUNS of consecutive records of at most ~256 MB into one reused buffer (the loader: pinned
    // memory of its own, gbits_buffer) - not whole: cutting, zero-filling and faulting in a buffer of the file's size cost more
    // than reading it.  Of a record of another shard only the head is read (names and contig table: rank 0 prints every shard's
    // rows), so a shard reads 1 / N of the genome bytes.
    std::vector<uint8_t> ib, own;
    size_t RUN_MAX = (size_t)256 << 20, HEAD = (size_t)256 << 10;
    if (const char *e = getenv("LM_LOADER_RUN_BYTES")) RUN_MAX = (size_t)std::max(1ll, atoll(e)); // (tests: runs of a few records, heads of a few bytes)
    if (const char *e = getenv("LM_LOADER_HEAD_BYTES")) HEAD = (size_t)std::max(1ll, atoll(e));
    for (int b = 0; b < out.genome_batches; b++) {
        char name[64];
        snprintf(name, sizeof name, "/genomes/batch_%04d/genomes.bin", b);
        if (!read_all(dir + name + ".idx", ib) || ib.size() < 24 || memcmp(ib.data(), ".genomei", 8) != 0) {
            status = ib.empty() ? 1 : 2;
            return std::string("genome data: invalid binary format: ") + name + ".idx";
        }
        uint32_t nrec = be32(&ib[20]);
        if (out.batch_first[b] != global) { // (set by load_index; the seed decoders read the table meanwhile)
            status = 2;
            return "genome data: the batch index changed while the index was being opened";
        }
        if (24 + (size_t)nrec * 12 > ib.size()) {
            status = 2;
            return "genome data: broken file (index)";
        }
        FdGuard fd(open((dir + name).c_str(), O_RDONLY));
        struct stat sb;
        uint8_t magic[16];
        if (fd.fd < 0 || fstat(fd.fd, &sb) != 0 || sb.st_size < 16 || !pread_all(fd.fd, magic, 16, 0) || memcmp(magic, ".genomes", 8) != 0 || magic[8] != 0) {
            status = fd.fd < 0 ? 1 : 2;
            return std::string("genome data: invalid binary format: ") + name;
        }
        const size_t fsize = (size_t)sb.st_size;
        auto rec_off = [&](uint32_t r) { return r < nrec ? (size_t)be64(&ib[24 + (size_t)r * 12]) : fsize; };
        for (uint32_t r0 = 0; r0 < nrec;) {
            const bool local0 = out.g2local.empty() || out.g2local[(size_t)(global)] >= 0;
            // the run: consecutive local records up to RUN_MAX bytes (at least one), or the head of ONE record of another shard
            uint32_t r1 = r0 + 1;
            size_t o0 = rec_off(r0), o1 = rec_off(r1);
            if (o0 < 16 || o1 < o0 || o1 > fsize) {
                status = 2;
                return "genome data: broken file (index)";
            }
            bool head_only = false;
            if (local0) {
                while (r1 < nrec && (out.g2local.empty() || out.g2local[(size_t)(global + (r1 - r0))] >= 0) && rec_off(r1 + 1) >= o1 &&
                       rec_off(r1 + 1) <= fsize && rec_off(r1 + 1) - o0 <= RUN_MAX)
                    o1 = rec_off(++r1);
            } else if (o1 - o0 > HEAD) {
                head_only = true;
            }
            size_t want = head_only ? HEAD : o1 - o0;
            uint8_t *gb = nullptr;
            for (int attempt = 0; attempt < 2; attempt++) { // (second attempt: the head of a foreign record was not all of its head)
                if (out.gbits_buffer) {
                    gb = out.gbits_buffer(want + 16);
                } else {
                    if (own.size() < want + 16) own.resize(want + 16);
                    gb = own.data();
                }
                const double tr0 = nowms();
                const bool read_ok = gb && pread_all_mt(fd.fd, gb, want, o0);
                t_read += nowms() - tr0;
                if (!read_ok) {
                    status = 1;
                    return std::string("genome data: cannot read ") + name;
                }
                const size_t gbn = want;
                bool again = false;
                const int64_t global0 = global, run0 = run;
                const size_t ng0 = out.genomes.size(), no0 = out.others.size();
                for (uint32_t r = r0; r < r1 && !again; r++, global++) {
                    const bool local = local0; // (a run is all local, or one foreign record)
                    size_t p = rec_off(r) - o0;
                    const size_t rend = head_only ? gbn : rec_off(r + 1) - o0;
                    HostGenome g;
                    g.bg = ((uint64_t)b << 17) | r;
                    g.global = global;
                    auto it = id_of.find(g.bg);
                    if (it != id_of.end()) g.id = it->second;
                    auto short_of = [&](size_t need) { // true: the record's bytes end before `need`
                        if (need <= rend) return false;
                        if (head_only) again = true; // (read the whole record)
                        return true;
                    };
                    if (short_of(p + 2)) break;
                    uint32_t idl = be16(&gb[p]);
                    p += 2 + idl;
                    if (short_of(p + 12)) break;
                    g.genome_size = (int32_t)be32(&gb[p]);
                    g.len = (int32_t)be32(&gb[p + 4]);
                    g.nseqs = (int32_t)be32(&gb[p + 8]);
                    p += 12;
                    bool cut = false;
                    for (int s = 0; s < g.nseqs; s++) {
                        if (short_of(p + 6)) {
                            cut = true;
                            break;
                        }
                        g.seq_sizes.push_back((int32_t)be32(&gb[p]));
                        uint32_t l = be16(&gb[p + 4]);
                        p += 6;
                        if (short_of(p + l)) {
                            cut = true;
                            break;
                        }
                        g.seq_ids.emplace_back((const char *)&gb[p], l);
                        p += l;
                    }
                    if (cut || short_of(p + 8)) break;
                    uint32_t nbytes = be32(&gb[p]);
                    p += 8;
                    if (!local) { // (its bases are another shard's: not read at all when the record is longer than its head)
                        if (o0 + p + nbytes > o1) {
                            status = 2;
                            return "genome data: broken file";
                        }
                        out.other_of[g.bg] = (int)out.others.size();
                        out.others.push_back(std::move(g));
                        continue;
                    }
                    if (p + nbytes > rend) {
                        status = 2;
                        return "genome data: broken file";
                    }
                    if ((int64_t)g.len > out.max_genome_len) {
                        // (the seed packer sized its position field from the .idx tables before this file was read: a record longer
                        // than its index entry says would have its positions packed into too few bits)
                        status = 2;
                        return std::string("genome data: a record of ") + name + " is longer than its genomes.bin.idx entry says";
                    }
                    if (to_sink) { // (same offsets as the host store: 8 .. 15 bytes of padding behind every genome)
                        g.bits_off = run;
                        const double ts0 = rdbg ? nowms() : 0;
                        const bool sunk = out.gbits_sink(gb + p, nbytes, run);
                        if (rdbg) t_sink += nowms() - ts0;
                        if (!sunk) {
                            status = 2;
                            return std::string("genome data: the packed bases of ") + name + " exceed what the batch indexes announced";
                        }
                        run = (run + (int64_t)nbytes + 15) & ~(int64_t)7;
                    } else {
                        g.bits_off = (int64_t)out.gbits.size();
                        out.gbits.insert(out.gbits.end(), gb + p, gb + p + nbytes);
                        // pad so that 8-byte loads near the end of a genome stay inside the buffer
                        size_t padded = (out.gbits.size() + 15) & ~(size_t)7;
                        out.gbits.resize(padded, 0);
                    }
                    out.genomes.push_back(std::move(g));
                }
                {
                    const double te0 = nowms();
                    if (out.gbits_batch_end) out.gbits_batch_end(); // (the buffer is read into again)
                    t_end += nowms() - te0;
                }
                if (!again) {
                    if (global != global0 + (int64_t)(r1 - r0)) { // a record ended before its fields did
                        status = 2;
                        return "genome data: broken file";
                    }
                    break;
                }
                if (attempt == 1) {
                    status = 2;
                    return "genome data: broken file";
                }
                global = global0; // the whole record this time
                run = run0;
                out.genomes.resize(ng0);
                out.others.resize(no0);
                head_only = false;
                want = o1 - o0;
            }
            r0 = r1;
        }
    }
    if (rdbg)
        fprintf(stderr, "[lm] genome reader: %.0f ms reading runs into the buffer, %.0f ms handing %zu records to the sink, %.0f ms waiting for the sink at the end of a run\n",
                t_read, t_sink, out.genomes.size(), t_end);
    (void)t_idx;
    out.gbits_total = to_sink ? run : (int64_t)out.gbits.size();

    return "";
}

} // namespace lm

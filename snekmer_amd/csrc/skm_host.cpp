// Host-only part of libsnekmer_hip.so: plain C++17, no HIP headers, so that the same file also builds with
// -fsanitize=address,undefined on the CPU (`make -C oracle asan`; tests/test_sanitizers.py drives it):
//
//   skm_last_error / skm_set_error       thread-local error text of the C ABI
//   skm_plan_alltoallv / _allgatherv     the byte plans of the two grouped RCCL exchanges (snekmer_amd/dist.py);
//                                        no reference counterpart (one process per FASTA file,
//                                        snekmer/rules/kmerize.smk:57-65)
//   skm_fasta_index / skm_fasta_parse    threaded FASTA reader that emits what the device entry points consume
//                                        (packed residues + offsets) plus the span of every record id; replaces the
//                                        `for f in SeqIO.parse(fasta, "fasta")` loops of rules/kmerize.smk:90-129
#include <cstring>
#include <thread>
#include <vector>

#include "skm_host.h"

static thread_local char g_err[1024] = "";

void skm_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *skm_last_error(void) { return g_err; }

// ------------------------------------------------------------------------------------------------ exchange plans
// A grouped exchange is ONE RCCL group of point-to-point transfers: every pair of ranks uses its own xGMI link and every
// piece lands at its final offset.  ops[p * narrays + a] = what this rank exchanges with peer p for array a.
extern "C" int skm_plan_alltoallv(int nranks, int narrays, const int64_t *h_elem_bytes, const int64_t *h_send_counts,
                                  const int64_t *h_recv_counts, skm_p2p_op *out_ops)
{
    SKM_REQUIRE(nranks >= 1 && nranks <= SKM_MAX_RANKS && narrays >= 1 && narrays <= SKM_MAX_ARRAYS && h_elem_bytes &&
                    h_send_counts && h_recv_counts && out_ops,
                SKM_E_BADARG, "skm_plan_alltoallv: bad argument");
    for (int a = 0; a < narrays; ++a) {
        SKM_REQUIRE(h_elem_bytes[a] > 0, SKM_E_BADARG, "skm_plan_alltoallv: element size of array %d", a);
        int64_t soff = 0, roff = 0;
        for (int p = 0; p < nranks; ++p) {
            SKM_REQUIRE(h_send_counts[p] >= 0 && h_recv_counts[p] >= 0, SKM_E_BADARG, "skm_plan_alltoallv: negative count for rank %d", p);
            skm_p2p_op &op = out_ops[p * narrays + a];
            op.peer = p;
            op.array = a;
            op.send_off = soff;
            op.send_bytes = h_send_counts[p] * h_elem_bytes[a];
            op.recv_off = roff;
            op.recv_bytes = h_recv_counts[p] * h_elem_bytes[a];
            soff += op.send_bytes;
            roff += op.recv_bytes;
        }
    }
    return SKM_OK;
}

extern "C" int skm_plan_allgatherv(int nranks, int rank, int narrays, const int64_t *h_elem_bytes, const int64_t *h_counts,
                                   skm_p2p_op *out_ops)
{
    SKM_REQUIRE(nranks >= 1 && nranks <= SKM_MAX_RANKS && rank >= 0 && rank < nranks && narrays >= 1 &&
                    narrays <= SKM_MAX_ARRAYS && h_elem_bytes && h_counts && out_ops,
                SKM_E_BADARG, "skm_plan_allgatherv: bad argument");
    for (int a = 0; a < narrays; ++a) {
        SKM_REQUIRE(h_elem_bytes[a] > 0, SKM_E_BADARG, "skm_plan_allgatherv: element size of array %d", a);
        int64_t roff = 0;
        for (int p = 0; p < nranks; ++p) {
            const int64_t cnt = h_counts[a * nranks + p];
            SKM_REQUIRE(cnt >= 0, SKM_E_BADARG, "skm_plan_allgatherv: negative count (array %d, rank %d)", a, p);
            skm_p2p_op &op = out_ops[p * narrays + a];
            op.peer = p;
            op.array = a;
            op.send_off = 0;  // every peer gets this rank's whole contribution
            op.send_bytes = h_counts[a * nranks + rank] * h_elem_bytes[a];
            op.recv_off = roff;
            op.recv_bytes = cnt * h_elem_bytes[a];
            roff += op.recv_bytes;
        }
    }
    return SKM_OK;
}


namespace {

// ------------------------------------------------------------------------------------------------ FASTA (host)
// Biopython (SimpleFastaParser, which SeqIO.parse(..., "fasta") drives) on a text-mode handle:
//   * lines end at "\n", "\r\n" or a lone "\r" (universal newlines);
//   * text before the first line that starts with '>' is skipped;
//   * a record's title is the header line without '>' and trailing whitespace; id = its first
//     whitespace-delimited word ("" if there is none);
//   * the sequence is the following lines, each with TRAILING whitespace removed, joined, then with every ' ' and
//     '\r' removed (leading tabs and interior characters stay).
// Bytes >= 0x80 would be decoded as UTF-8 by the text handle: they are reported (flag bit 0) and the caller takes
// its text-mode path; this parser is for the ASCII files FASTA is.
inline bool is_space(uint8_t c)  // str.isspace() over ASCII
{
    return c == ' ' || (c >= 9 && c <= 13) || (c >= 0x1c && c <= 0x1f);
}

struct chunk_stat {
    int64_t begin = 0, end = 0;  // [begin, end) of the buffer, both at line starts
    int64_t nrec = 0;            // header lines inside
    int64_t nres = 0;            // sequence bytes inside (after the line rules)
    int64_t pre = 0;             // of those, the ones before the chunk's first header (they continue an earlier record)
    int nonascii = 0;
};

inline int64_t line_end(const uint8_t *buf, int64_t p, int64_t end)
{
    // two vectorised searches instead of a byte loop: the next "\n", then a "\r" in front of it (rare)
    const void *nl = memchr(buf + p, '\n', (size_t)(end - p));
    const int64_t stop = nl ? (const uint8_t *)nl - buf : end;
    const void *cr = memchr(buf + p, '\r', (size_t)(stop - p));
    return cr ? (const uint8_t *)cr - buf : stop;
}

inline int64_t next_line(const uint8_t *buf, int64_t le, int64_t len)
{
    if (le >= len)
        return len;
    if (buf[le] == '\r' && le + 1 < len && buf[le + 1] == '\n')
        return le + 2;
    return le + 1;
}

// first line start at or after p
inline int64_t align_to_line(const uint8_t *buf, int64_t p, int64_t len)
{
    if (p <= 0)
        return 0;
    if (p >= len)
        return len;
    // p is a line start iff the previous byte ends a line ("\r\n" counts once)
    for (;; ++p) {
        if (p >= len)
            return len;
        const uint8_t prev = buf[p - 1];
        if (prev == '\n' || (prev == '\r' && buf[p] != '\n'))
            return p;
    }
}

// bytes a sequence line contributes; when `out` is given they are written there
inline int64_t emit_sequence_line(const uint8_t *buf, int64_t ls, int64_t le, uint8_t *out)
{
    while (le > ls && is_space(buf[le - 1]))
        --le;
    // a line never holds "\r" (it ends there); one without a space is copied whole
    if (!memchr(buf + ls, ' ', (size_t)(le - ls))) {
        if (out)
            memcpy(out, buf + ls, (size_t)(le - ls));
        return le - ls;
    }
    int64_t w = 0;
    for (int64_t p = ls; p < le; ++p) {
        const uint8_t c = buf[p];
        if (c == ' ')
            continue;
        if (out)
            out[w] = c;
        ++w;
    }
    return w;
}

inline int any_high_bit(const uint8_t *buf, int64_t b, int64_t e)
{
    uint64_t acc = 0;
    int64_t p = b;
    for (; p + 8 <= e; p += 8) {
        uint64_t w;
        memcpy(&w, buf + p, 8);
        acc |= w;
    }
    for (; p < e; ++p)
        acc |= (uint64_t)buf[p];
    return (acc & 0x8080808080808080ull) ? 1 : 0;
}

void scan_chunk(const uint8_t *buf, int64_t len, chunk_stat *cs)
{
    bool seen_header = false;
    const int nonascii = any_high_bit(buf, cs->begin, cs->end);
    for (int64_t ls = cs->begin; ls < cs->end;) {
        const int64_t le = line_end(buf, ls, len);
        if (le > ls && buf[ls] == '>') {
            seen_header = true;
            ++cs->nrec;
        } else {
            const int64_t w = emit_sequence_line(buf, ls, le, nullptr);
            cs->nres += w;
            if (!seen_header)
                cs->pre += w;
        }
        ls = next_line(buf, le, len);
    }
    cs->nonascii = nonascii;
}

struct chunk_base {
    int64_t rec = 0;   // index of the chunk's first record
    int64_t res = 0;   // residue position where the chunk's first kept byte lands
    bool open = false;  // a record is open when the chunk starts (its leading sequence lines are kept)
};

void write_chunk(const uint8_t *buf, int64_t len, const chunk_stat *cs, chunk_base base, uint8_t *out_res,
                 int64_t *out_off, int64_t *id_begin, int32_t *id_len)
{
    bool open = base.open;
    int64_t rec = base.rec, pos = base.res;
    for (int64_t ls = cs->begin; ls < cs->end;) {
        const int64_t le = line_end(buf, ls, len);
        if (le > ls && buf[ls] == '>') {
            int64_t b = ls + 1;
            while (b < le && is_space(buf[b]))
                ++b;
            int64_t e = b;
            while (e < le && !is_space(buf[e]))
                ++e;
            out_off[rec] = pos;
            id_begin[rec] = b;
            id_len[rec] = (int32_t)(e - b);
            ++rec;
            open = true;
        } else if (open) {
            pos += emit_sequence_line(buf, ls, le, out_res + pos);
        }
        ls = next_line(buf, le, len);
    }
}

int plan_chunks(const uint8_t *buf, int64_t len, int nthreads, std::vector<chunk_stat> &chunks)
{
    if (nthreads < 1)
        nthreads = (int)std::thread::hardware_concurrency();
    if (nthreads < 1)
        nthreads = 1;
    if (nthreads > 64)
        nthreads = 64;
    const int64_t min_chunk = 1 << 20;  // below a megabyte per thread the spawn costs more than the scan
    int64_t parts = len / min_chunk;
    if (parts < 1)
        parts = 1;
    if (parts > nthreads)
        parts = nthreads;
    chunks.assign((size_t)parts, chunk_stat());
    int64_t prev = 0;
    for (int64_t t = 0; t < parts; ++t) {
        chunks[t].begin = prev;
        const int64_t want = t + 1 == parts ? len : align_to_line(buf, len / parts * (t + 1), len);
        chunks[t].end = want < prev ? prev : want;
        prev = chunks[t].end;
    }
    return (int)parts;
}

template <typename F>
void run_chunks(int parts, F &&fn)
{
    if (parts == 1) {
        fn(0);
        return;
    }
    std::vector<std::thread> th;
    th.reserve((size_t)parts);
    for (int t = 0; t < parts; ++t)
        th.emplace_back([&fn, t] { fn(t); });
    for (auto &x : th)
        x.join();
}

}  // namespace

extern "C" int skm_fasta_index(const uint8_t *h_buf, int64_t len, int nthreads, int64_t *out_nrecords,
                               int64_t *out_nresidues, int *out_flags)
{
    SKM_REQUIRE(len >= 0 && (len == 0 || h_buf) && out_nrecords && out_nresidues && out_flags, SKM_E_BADARG,
                "skm_fasta_index: bad argument");
    std::vector<chunk_stat> chunks;
    const int parts = plan_chunks(h_buf, len, nthreads, chunks);
    run_chunks(parts, [&](int t) { scan_chunk(h_buf, len, &chunks[(size_t)t]); });
    int64_t nrec = 0, nres = 0;
    int flags = 0;
    for (const chunk_stat &c : chunks) {
        nres += nrec ? c.nres : c.nres - c.pre;  // sequence text before the file's first header is skipped
        nrec += c.nrec;
        flags |= c.nonascii ? 1 : 0;
    }
    *out_nrecords = nrec;
    *out_nresidues = nres;
    *out_flags = flags;
    return SKM_OK;
}

extern "C" int skm_fasta_parse(const uint8_t *h_buf, int64_t len, int nthreads, int64_t nrecords, int64_t nresidues,
                               uint8_t *h_out_residues, int64_t *h_out_offsets, int64_t *h_out_id_begin,
                               int32_t *h_out_id_len)
{
    SKM_REQUIRE(len >= 0 && (len == 0 || h_buf) && nrecords >= 0 && nresidues >= 0 && h_out_offsets, SKM_E_BADARG,
                "skm_fasta_parse: bad argument");
    SKM_REQUIRE((nresidues == 0 || h_out_residues) && (nrecords == 0 || (h_out_id_begin && h_out_id_len)), SKM_E_BADARG,
                "skm_fasta_parse: null output array");
    std::vector<chunk_stat> chunks;
    const int parts = plan_chunks(h_buf, len, nthreads, chunks);
    run_chunks(parts, [&](int t) { scan_chunk(h_buf, len, &chunks[(size_t)t]); });
    std::vector<chunk_base> base((size_t)parts);
    int64_t nrec = 0, nres = 0;
    for (int t = 0; t < parts; ++t) {
        base[(size_t)t].rec = nrec;
        base[(size_t)t].res = nres;
        base[(size_t)t].open = nrec > 0;
        nres += nrec ? chunks[(size_t)t].nres : chunks[(size_t)t].nres - chunks[(size_t)t].pre;
        nrec += chunks[(size_t)t].nrec;
    }
    SKM_REQUIRE(nrec == nrecords && nres == nresidues, SKM_E_BADARG,
                "skm_fasta_parse: the buffer holds %lld records / %lld residues, the caller sized for %lld / %lld",
                (long long)nrec, (long long)nres, (long long)nrecords, (long long)nresidues);
    run_chunks(parts, [&](int t) {
        write_chunk(h_buf, len, &chunks[(size_t)t], base[(size_t)t], h_out_residues, h_out_offsets, h_out_id_begin, h_out_id_len);
    });
    h_out_offsets[nrecords] = nresidues;
    return SKM_OK;
}


// ------------------------------------------------------------------------------------------------ .npz writer
// np.savez_compressed (what rules/kmerize.smk:132-139 calls) deflates every member on one thread: 15 s for the
// 0.96 GB of arrays of a 100 k-sequence file.  Here every member's bytes (its .npy header, then the array) are cut
// into chunks that a pool of threads deflates independently, pigz-style: each chunk is a run of raw-deflate blocks
// closed by a sync flush (byte-aligned, not final), the last chunk of a member is finished, so their concatenation is
// ONE valid raw-deflate stream; the member's CRC-32 is combined from the chunks' CRCs.  The result is an ordinary zip
// archive (zip64 records as numpy itself writes them) that np.load / snekmer.io.load_npz read unchanged.
#include <atomic>
#include <cstdlib>
#include <mutex>
#include <string>

#include <sched.h>
#include <zlib.h>

namespace {

struct npz_chunk {
    int member;
    const uint8_t *src;
    size_t len;
    bool last;
    std::vector<uint8_t> out;
    uint32_t crc = 0;
    int err = Z_OK;
};

void put16(std::vector<uint8_t> &b, uint32_t v)
{
    b.push_back((uint8_t)v);
    b.push_back((uint8_t)(v >> 8));
}
void put32(std::vector<uint8_t> &b, uint32_t v)
{
    put16(b, v & 0xFFFFu);
    put16(b, v >> 16);
}
void put64(std::vector<uint8_t> &b, uint64_t v)
{
    put32(b, (uint32_t)v);
    put32(b, (uint32_t)(v >> 32));
}

void npz_deflate_chunk(npz_chunk &c, int level)
{
    c.crc = (uint32_t)crc32(0L, Z_NULL, 0);
    for (size_t at = 0; at < c.len;) {  // crc32 takes 32-bit lengths
        const size_t piece = c.len - at < ((size_t)1 << 30) ? c.len - at : ((size_t)1 << 30);
        c.crc = (uint32_t)crc32(c.crc, c.src + at, (uInt)piece);
        at += piece;
    }
    if (level == 0) {  // stored member: the bytes are written from the caller's buffer
        return;
    }
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    c.err = deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
    if (c.err != Z_OK)
        return;
    c.out.resize(deflateBound(&zs, (uLong)c.len) + 16);
    zs.next_in = const_cast<Bytef *>(c.src);
    zs.avail_in = (uInt)c.len;
    zs.next_out = c.out.data();
    zs.avail_out = (uInt)c.out.size();
    const int rc = deflate(&zs, c.last ? Z_FINISH : Z_SYNC_FLUSH);
    if ((c.last && rc != Z_STREAM_END) || (!c.last && (rc != Z_OK || zs.avail_in != 0)))
        c.err = rc == Z_OK ? Z_BUF_ERROR : rc;
    c.out.resize(zs.total_out);
    deflateEnd(&zs);
}

}  // namespace

extern "C" int skm_npz_write(const char *path, int nmembers, const char *const *names, const void *const *h_headers,
                             const int64_t *header_bytes, const void *const *h_data, const int64_t *data_bytes, int level,
                             int nthreads, int64_t *out_file_bytes)
{
    SKM_REQUIRE(path && nmembers >= 0 && nmembers <= 65535 && (nmembers == 0 || (names && h_headers && header_bytes && h_data && data_bytes)),
                SKM_E_BADARG, "skm_npz_write: bad argument");
    SKM_REQUIRE(level >= -1 && level <= 9, SKM_E_BADARG, "skm_npz_write: level must be -1 (zlib's default) or 0 (stored) to 9");
    if (level < 0)
        level = 6;
    constexpr size_t CHUNK = (size_t)1 << 21;  // 2 MiB: ~64 back-reference windows, a few ms of deflate
    std::vector<npz_chunk> chunks;
    std::vector<size_t> first_chunk((size_t)nmembers + 1, 0);
    for (int m = 0; m < nmembers; ++m) {
        SKM_REQUIRE(names[m] && strlen(names[m]) > 0 && strlen(names[m]) < 60000 && header_bytes[m] > 0 && h_headers[m] &&
                        data_bytes[m] >= 0 && (data_bytes[m] == 0 || h_data[m]),
                    SKM_E_BADARG, "skm_npz_write: bad member %d", m);
        first_chunk[m] = chunks.size();
        const bool empty = data_bytes[m] == 0;
        chunks.push_back({m, (const uint8_t *)h_headers[m], (size_t)header_bytes[m], empty, {}, 0, Z_OK});
        for (size_t at = 0; at < (size_t)data_bytes[m]; at += CHUNK) {
            const size_t len = (size_t)data_bytes[m] - at < CHUNK ? (size_t)data_bytes[m] - at : CHUNK;
            chunks.push_back({m, (const uint8_t *)h_data[m] + at, len, at + len == (size_t)data_bytes[m], {}, 0, Z_OK});
        }
    }
    first_chunk[nmembers] = chunks.size();
    unsigned hw = std::thread::hardware_concurrency();
    {
        cpu_set_t set;  // the cores this process may run on (a cgroup / taskset share of the host), not the host's
        if (sched_getaffinity(0, sizeof(set), &set) == 0 && CPU_COUNT(&set) > 0)
            hw = (unsigned)CPU_COUNT(&set);
    }
    size_t nt = nthreads >= 1 ? (size_t)nthreads : (hw ? hw : 1);
    if (nt > 64)
        nt = 64;
    if (nt > chunks.size())
        nt = chunks.size() ? chunks.size() : 1;
    std::atomic<size_t> next{0};
    auto work = [&]() {
        for (size_t i = next.fetch_add(1); i < chunks.size(); i = next.fetch_add(1))
            npz_deflate_chunk(chunks[i], level);
    };
    {
        std::vector<std::thread> pool;
        for (size_t t = 1; t < nt; ++t)
            pool.emplace_back(work);
        work();
        for (auto &th : pool)
            th.join();
    }
    for (auto &c : chunks)
        SKM_REQUIRE(c.err == Z_OK, SKM_E_UNSUPPORTED, "skm_npz_write: zlib error %d in member %s", c.err, names[c.member]);

    FILE *f = fopen(path, "wb");
    SKM_REQUIRE(f, SKM_E_BADARG, "skm_npz_write: cannot open %s for writing", path);
    uint64_t offset = 0;
    bool ok = true;
    auto emit = [&](const void *p, size_t nbytes) {
        if (nbytes && fwrite(p, 1, nbytes, f) != nbytes)
            ok = false;
        offset += nbytes;
    };
    std::vector<uint8_t> central;
    const uint32_t method = level == 0 ? 0u : 8u;
    const uint32_t dos_time = 0, dos_date = (1u << 5) | 1u;  // 1980-01-01 00:00, what zipfile writes for "no date"
    for (int m = 0; m < nmembers; ++m) {
        uint64_t raw = 0, packed = 0;
        uint32_t crc = (uint32_t)crc32(0L, Z_NULL, 0);
        for (size_t i = first_chunk[m]; i < first_chunk[m + 1]; ++i) {
            const npz_chunk &c = chunks[i];
            crc = (uint32_t)crc32_combine(crc, c.crc, (z_off_t)c.len);
            raw += c.len;
            packed += level == 0 ? c.len : c.out.size();
        }
        const std::string fname = std::string(names[m]) + ".npy";
        const uint64_t local_at = offset;
        std::vector<uint8_t> h;
        put32(h, 0x04034b50u);
        put16(h, 45);  // zip64
        put16(h, 0);
        put16(h, method);
        put16(h, dos_time);
        put16(h, dos_date);
        put32(h, crc);
        put32(h, 0xFFFFFFFFu);
        put32(h, 0xFFFFFFFFu);
        put16(h, (uint32_t)fname.size());
        put16(h, 20);
        h.insert(h.end(), fname.begin(), fname.end());
        put16(h, 0x0001);
        put16(h, 16);
        put64(h, raw);
        put64(h, packed);
        emit(h.data(), h.size());
        for (size_t i = first_chunk[m]; i < first_chunk[m + 1]; ++i) {
            npz_chunk &c = chunks[i];
            if (level == 0)
                emit(c.src, c.len);
            else
                emit(c.out.data(), c.out.size());
            std::vector<uint8_t>().swap(c.out);
        }
        put32(central, 0x02014b50u);
        put16(central, (3u << 8) | 45u);
        put16(central, 45);
        put16(central, 0);
        put16(central, method);
        put16(central, dos_time);
        put16(central, dos_date);
        put32(central, crc);
        put32(central, 0xFFFFFFFFu);
        put32(central, 0xFFFFFFFFu);
        put16(central, (uint32_t)fname.size());
        put16(central, 28);
        put16(central, 0);
        put16(central, 0);
        put16(central, 0);
        put32(central, 0600u << 16);
        put32(central, 0xFFFFFFFFu);
        central.insert(central.end(), fname.begin(), fname.end());
        put16(central, 0x0001);
        put16(central, 24);
        put64(central, raw);
        put64(central, packed);
        put64(central, local_at);
    }
    const uint64_t cd_at = offset, cd_size = central.size();
    emit(central.data(), central.size());
    std::vector<uint8_t> tail;
    const uint64_t eocd64_at = offset;
    put32(tail, 0x06064b50u);
    put64(tail, 44);
    put16(tail, (3u << 8) | 45u);
    put16(tail, 45);
    put32(tail, 0);
    put32(tail, 0);
    put64(tail, (uint64_t)nmembers);
    put64(tail, (uint64_t)nmembers);
    put64(tail, cd_size);
    put64(tail, cd_at);
    put32(tail, 0x07064b50u);
    put32(tail, 0);
    put64(tail, eocd64_at);
    put32(tail, 1);
    put32(tail, 0x06054b50u);
    put16(tail, 0);
    put16(tail, 0);
    put16(tail, (uint32_t)nmembers);
    put16(tail, (uint32_t)nmembers);
    put32(tail, cd_size < 0xFFFFFFFFull ? (uint32_t)cd_size : 0xFFFFFFFFu);
    put32(tail, cd_at < 0xFFFFFFFFull ? (uint32_t)cd_at : 0xFFFFFFFFu);
    put16(tail, 0);
    emit(tail.data(), tail.size());
    if (fclose(f) != 0)
        ok = false;
    SKM_REQUIRE(ok, SKM_E_UNSUPPORTED, "skm_npz_write: short write to %s", path);
    if (out_file_bytes)
        *out_file_bytes = (int64_t)offset;
    return SKM_OK;
}

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
#include <algorithm>
#include <cmath>
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
    int utf32_item = 0;  // > 0: the bytes are little-endian UTF-32 strings ('<U' arrays) of this many bytes per item;
                         // < 0: little-endian 4- or 8-byte numbers (the same encoder on 4-byte units, no item structure)
};

// ---------------------------------------------------------------------------------------------------------------------
// A deflate encoder for numpy '<U' arrays (UTF-32: kmerlist, seqs, ids - three quarters of a k-mer file's bytes).
// zlib spends its time in hash chains there: every position of "c 0 0 0 c 0 0 0 ..." hashes like thousands of earlier
// ones, and at level 6 it walks 128 of them per byte (16-22 MB/s per thread).  The structure is known, so no search is
// needed: a 4-byte unit either repeats the unit at the same place of the previous item (sorted k-mers share long
// prefixes: one match of distance item-size), or repeats the previous unit (zero padding), or is one literal followed
// by a 3-byte match at distance 4 (the three zero bytes of every character below U+0100).  The tokens are coded with
// dynamic Huffman trees built per chunk (RFC 1951 section 3.2.7); any inflate reads the result.
namespace utf32_deflate {

struct bit_writer {
    std::vector<uint8_t> &out;
    uint64_t acc = 0;
    int nbits = 0;
    explicit bit_writer(std::vector<uint8_t> &o) : out(o) {}
    void put(uint32_t value, int n)  // n <= 32, least significant bit first
    {
        acc |= (uint64_t)value << nbits;
        nbits += n;
        while (nbits >= 8) {
            out.push_back((uint8_t)acc);
            acc >>= 8;
            nbits -= 8;
        }
    }
    void align()
    {
        if (nbits > 0) {
            out.push_back((uint8_t)acc);
            acc = 0;
            nbits = 0;
        }
    }
};

// Code lengths (<= maxbits) of a Huffman code for freq[0..n): a plain Huffman tree, then the usual repair of the
// length histogram when the tree is deeper than maxbits.  At least two symbols get a code (a lone symbol's partner is
// the unused symbol next to it), so the code is always complete.
void code_lengths(const uint32_t *freq, int n, int maxbits, uint8_t *len)
{
    std::vector<int> used;
    for (int i = 0; i < n; ++i) {
        len[i] = 0;
        if (freq[i])
            used.push_back(i);
    }
    if (used.empty())
        used.push_back(0);
    if (used.size() == 1)
        used.push_back(used[0] == 0 ? 1 : used[0] - 1);
    std::sort(used.begin(), used.end(), [&](int a, int b) { return freq[a] != freq[b] ? freq[a] < freq[b] : a < b; });
    const int m = (int)used.size();
    // two-queue construction over the sorted leaves; node 0..m-1 leaves, m.. internal
    std::vector<uint64_t> w((size_t)2 * m);
    std::vector<int> parent((size_t)2 * m, -1);
    for (int i = 0; i < m; ++i)
        w[i] = freq[used[i]] ? freq[used[i]] : 1;
    int leaf = 0, inner = m, next = m;
    auto take = [&]() {
        if (leaf < m && (inner >= next || w[leaf] <= w[inner]))
            return leaf++;
        return inner++;
    };
    while (next < 2 * m - 1) {
        const int a = take(), b = take();
        w[next] = w[a] + w[b];
        parent[a] = parent[b] = next;
        ++next;
    }
    std::vector<int> count((size_t)maxbits + 64, 0);
    std::vector<int> depth((size_t)2 * m, 0);
    for (int i = 2 * m - 3; i >= 0; --i)
        depth[i] = depth[parent[i]] + 1;
    int deepest = 0;
    for (int i = 0; i < m; ++i) {
        const int d = depth[i] < maxbits + 60 ? depth[i] : maxbits + 60;
        ++count[d];
        deepest = d > deepest ? d : deepest;
    }
    if (deepest > maxbits) {
        for (int d = maxbits + 1; d <= deepest; ++d) {
            count[maxbits] += count[d];
            count[d] = 0;
        }
        uint64_t total = 0;
        for (int d = maxbits; d >= 1; --d)
            total += (uint64_t)count[d] << (maxbits - d);
        while (total != ((uint64_t)1 << maxbits)) {  // over-subscribed: lengthen one short code, pair up a longest one
            --count[maxbits];
            for (int d = maxbits - 1; d >= 1; --d)
                if (count[d]) {
                    --count[d];
                    count[d + 1] += 2;
                    break;
                }
            --total;
        }
    }
    // the rarest symbols get the longest codes
    int at = 0;
    for (int d = maxbits; d >= 1; --d)
        for (int c = 0; c < count[d]; ++c)
            len[used[at++]] = (uint8_t)d;
}

void canonical_codes(const uint8_t *len, int n, uint16_t *code)
{
    int bl[16] = {};
    for (int i = 0; i < n; ++i)
        ++bl[len[i]];
    bl[0] = 0;
    uint32_t next[16] = {};
    uint32_t c = 0;
    for (int b = 1; b < 16; ++b) {
        c = (c + (uint32_t)bl[b - 1]) << 1;
        next[b] = c;
    }
    for (int i = 0; i < n; ++i) {
        if (!len[i]) {
            code[i] = 0;
            continue;
        }
        uint32_t v = next[len[i]]++, r = 0;  // Huffman codes go out most significant bit first
        for (int b = 0; b < len[i]; ++b)
            r |= ((v >> b) & 1u) << (len[i] - 1 - b);
        code[i] = (uint16_t)r;
    }
}

const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

struct len_table {
    uint8_t sym[259];
    len_table()
    {
        for (int length = 0; length < 259; ++length) {
            int s = 28;
            while (s > 0 && LEN_BASE[s] > length)
                --s;
            sym[length] = (uint8_t)s;
        }
    }
};
inline int len_symbol(int length)  // 3..258 -> 0..28
{
    static const len_table t;
    return t.sym[length];
}
inline int dist_symbol(int dist)  // 1..32768 -> 0..29
{
    if (dist <= 4)
        return dist - 1;
    const int top = 31 - __builtin_clz((unsigned)(dist - 1));  // 2^top <= dist - 1 < 2^(top + 1)
    return 2 * top + (((dist - 1) >> (top - 1)) & 1);
}

// token: literal byte, or 0x80000000 | length << 16 | (distance - 1)
inline uint32_t match_token(int length, int dist) { return 0x80000000u | ((uint32_t)length << 16) | (uint32_t)(dist - 1); }

// One chunk (a whole number of 4-byte units; a whole number of items when item_bytes > 0) -> raw deflate: one dynamic
// block, then an empty stored block that byte-aligns the stream (what Z_SYNC_FLUSH leaves) or, for the member's last
// chunk, closes it.
void encode(const uint8_t *src, size_t len, int item_bytes, bool last, std::vector<uint8_t> &out)
{
    const size_t units = len / 4;
    const size_t w = item_bytes > 0 && item_bytes <= 32768 ? (size_t)item_bytes / 4 : 0;  // the window reaches one item back
    auto unit = [&](size_t i) {
        uint32_t v;
        memcpy(&v, src + 4 * i, 4);
        return v;
    };
    // units from `from` on that equal the unit `back` units before them (two units per step while both match)
    auto run_back = [&](size_t from, size_t back) {
        size_t r = 0;
        while (from + r + 2 <= units) {
            uint64_t a, b;
            memcpy(&a, src + 4 * (from + r), 8);
            memcpy(&b, src + 4 * (from + r - back), 8);
            if (a != b)
                break;
            r += 2;
        }
        while (from + r < units && unit(from + r) == unit(from + r - back))
            ++r;
        return r;
    };
    std::vector<uint32_t> tok;
    tok.reserve(units + 16);
    uint32_t flit[286] = {}, fdist[30] = {};
    // only two distances occur: the previous unit (4) and the previous item
    const int ds_unit = dist_symbol(4);
    auto emit_bytes = [&](size_t nbytes, int dist) {  // one match of nbytes >= 3 bytes, cut into pieces of 3..258
        const int ds = dist == 4 ? ds_unit : dist_symbol(dist);
        while (nbytes) {
            size_t piece = nbytes < 258 ? nbytes : 258;  // (258 has the cheapest length code: no extra bits)
            if (nbytes - piece > 0 && nbytes - piece < 3)
                piece = nbytes - 3;
            tok.push_back(match_token((int)piece, dist));
            ++flit[257 + len_symbol((int)piece)];
            ++fdist[ds];
            nbytes -= piece;
        }
    };
    auto emit_match = [&](size_t nunits, int dist) { emit_bytes(nunits * 4, dist); };
    // third candidate, for items longer than 32 characters: the last place the same PAIR of units stood (one table
    // entry per pair hash, no chains): free text (protein sequences) repeats pairs and triples of characters within the
    // 8192 characters the window holds.
    // A far match pays ~4 + log2(distance in units) bits, a unit coded alone ~3 + log2(distinct byte values): the match
    // is taken when it is the cheaper way to code its units.
    constexpr int HBITS = 15;
    std::vector<uint32_t> head, head_wide;
    // and, before it, the last place the same RUN of `wide` units stood, `wide` chosen so that a run is unlikely to
    // recur by chance inside the window (6 units for a 6-letter alphabet, 4 for 20 letters): in a file of protein
    // families that is the same stretch of a relative a few records back, and the match runs on for dozens of units.
    size_t wide = 4;
    auto pair_hash = [&](size_t i) { return (uint32_t)(((uint64_t)unit(i) * 0x9E3779B1u + (uint64_t)unit(i + 1) * 0x85EBCA77u) >> 7) & ((1u << HBITS) - 1u); };
    double unit_bits = 7.0;
    // (measured: free text - protein sequences, 20 letters - shrinks 9 % with the pair matches; sorted k-mer lists of
    // every alphabet and id lists grow 1-8 % with them, because the frequent short codes get longer: short items go without)
    const bool far_ok = item_bytes <= 0 || item_bytes > 4 * 32;
    {
        bool seen[256] = {};
        int distinct = 0;
        for (size_t i = 0; i < units && i < 4096; ++i) {
            const uint8_t b = src[4 * i];
            distinct += !seen[b];
            seen[b] = true;
        }
        unit_bits = 3.0 + log2((double)(distinct > 1 ? distinct : 2));
        wide = (size_t)ceil(15.0 / log2((double)(distinct > 1 ? distinct : 2)));
        wide = wide < 3 ? 3 : (wide > 8 ? 8 : wide);
    }
    if (far_ok) {
        head.assign((size_t)1 << HBITS, 0xFFFFFFFFu);
        head_wide.assign((size_t)1 << HBITS, 0xFFFFFFFFu);
    }
    auto wide_hash = [&](size_t i) {
        uint64_t h = 0;
        for (size_t j = 0; j < wide; ++j)
            h = (h + unit(i + j)) * 0x9E3779B97F4A7C15ull;
        return (uint32_t)(h >> (64 - HBITS));
    };
    auto note = [&](size_t from, size_t count) {  // positions a token covered enter the tables
        if (!far_ok)
            return;
        for (size_t i = from; i < from + count && i + 1 < units; ++i) {
            head[pair_hash(i)] = (uint32_t)i;
            if (i + wide <= units)
                head_wide[wide_hash(i)] = (uint32_t)i;
        }
    };
    for (size_t p = 0; p < units;) {
        const uint32_t u = unit(p);
        size_t r_item = 0, r_unit = 0, r_far = 0, far_at = 0, r_shift = 0, r_two = 0;
        if (w && p >= w)
            r_item = run_back(p, w);
        if (p >= 1 && u == unit(p - 1))
            r_unit = run_back(p, 1);
        if (w == 0 && p >= 2 && r_unit == 0 && u == unit(p - 2))  // numbers: the unit two back (8-byte elements)
            r_two = run_back(p, 2);
        if (w > 2 && p >= w - 1 && r_item < w - 1)  // the previous item moved up by one character (k-mers in window order)
            r_shift = run_back(p, w - 1);
        if (r_shift > r_item && r_shift > r_unit && r_shift >= 2) {
            emit_match(r_shift, (int)(w - 1) * 4);
            note(p, r_shift);
            p += r_shift;
            continue;
        }
        if (far_ok && r_item < wide && r_unit < wide && p + wide <= units) {
            const uint32_t q = head_wide[wide_hash(p)];
            if (q != 0xFFFFFFFFu && p - q <= 8192) {
                size_t r = 0;
                while (r < 64 && p + r < units && unit(q + r) == unit(p + r))
                    ++r;
                if (r >= wide) {
                    r_far = r;
                    far_at = q;
                }
            }
        }
        if (far_ok && r_far == 0 && r_item < 2 && r_unit < 2 && p + 1 < units) {
            const uint32_t q = head[pair_hash(p)];
            if (q != 0xFFFFFFFFu && p - q <= 8192 && unit(q) == u && unit(q + 1) == unit(p + 1)) {
                r_far = 2;
                while (r_far < 64 && p + r_far < units && unit(q + r_far) == unit(p + r_far))
                    ++r_far;
                far_at = q;
                if (4.0 + log2((double)(p - q)) + 2.0 >= unit_bits * (double)r_far)
                    r_far = 0;
            }
        }
        if (r_two > r_unit && r_two >= r_far && r_two >= r_item) {
            emit_match(r_two, 8);
            note(p, r_two);
            p += r_two;
        } else if (r_unit >= r_item && r_unit >= r_far && r_unit > 0) {
            emit_match(r_unit, 4);
            note(p, r_unit);
            p += r_unit;
        } else if (r_item >= r_far && r_item > 0) {
            emit_match(r_item, (int)w * 4);
            note(p, r_item);
            p += r_item;
        } else if (r_far > 0) {
            const int dist = (int)(p - far_at) * 4;
            tok.push_back(match_token((int)r_far * 4, dist));
            ++flit[257 + len_symbol((int)r_far * 4)];
            ++fdist[dist_symbol(dist)];
            note(p, r_far);
            p += r_far;
        } else {
            tok.push_back(u & 0xFFu);
            ++flit[u & 0xFFu];
            // the unit's other three bytes, and whole units behind them, copied from one unit back (characters below
            // U+0100, small integers) or - numbers only - two units back (8-byte elements that differ in the low byte)
            size_t best = 0;
            int best_dist = 0;
            for (int back = 1; back <= (w == 0 ? 8 : 1) && best == 0; ++back) {  // (numbers: the nearest of the last eight)
                if (p < (size_t)back || (u >> 8) != (unit(p - back) >> 8))
                    continue;
                size_t more = 0;
                while (w == 0 && more < 60 && p + 1 + more < units && unit(p + 1 + more) == unit(p + 1 + more - back))
                    ++more;
                if (1 + more > best) {
                    best = 1 + more;
                    best_dist = 4 * back;
                }
            }
            if (best) {
                emit_bytes(3 + 4 * (best - 1), best_dist);
                note(p, best);
                p += best;
            } else {
                note(p, 1);
                for (int b = 1; b < 4; ++b) {
                    tok.push_back((u >> (8 * b)) & 0xFFu);
                    ++flit[(u >> (8 * b)) & 0xFFu];
                }
                ++p;
            }
        }
    }
    for (size_t i = units * 4; i < len; ++i) {  // (a '<U' array is whole units; kept for safety)
        tok.push_back(src[i]);
        ++flit[src[i]];
    }
    ++flit[256];
    uint8_t llen[286], dlen[30];
    uint16_t lcode[286], dcode[30];
    code_lengths(flit, 286, 15, llen);
    code_lengths(fdist, 30, 15, dlen);
    canonical_codes(llen, 286, lcode);
    canonical_codes(dlen, 30, dcode);
    int hlit = 286, hdist = 30;
    while (hlit > 257 && llen[hlit - 1] == 0)
        --hlit;
    while (hdist > 1 && dlen[hdist - 1] == 0)
        --hdist;
    // the two length vectors, run-length coded with the symbols 16 / 17 / 18
    std::vector<uint8_t> seq(llen, llen + hlit);
    seq.insert(seq.end(), dlen, dlen + hdist);
    std::vector<uint16_t> cl;  // symbol | extra << 8
    uint32_t fcl[19] = {};
    for (size_t i = 0; i < seq.size();) {
        size_t run = 1;
        while (i + run < seq.size() && seq[i + run] == seq[i])
            ++run;
        const uint8_t v = seq[i];
        size_t left = run;
        if (v == 0) {
            while (left >= 3) {
                const size_t take = left < 138 ? left : 138;
                if (take >= 11) {
                    cl.push_back((uint16_t)(18 | ((take - 11) << 8)));
                    ++fcl[18];
                } else {
                    cl.push_back((uint16_t)(17 | ((take - 3) << 8)));
                    ++fcl[17];
                }
                left -= take;
            }
        } else {
            cl.push_back(v);
            ++fcl[v];
            --left;
            while (left >= 3) {
                const size_t take = left < 6 ? left : 6;
                cl.push_back((uint16_t)(16 | ((take - 3) << 8)));
                ++fcl[16];
                left -= take;
            }
        }
        for (; left; --left) {
            cl.push_back(v);
            ++fcl[v];
        }
        i += run;
    }
    uint8_t cllen[19];
    uint16_t clcode[19];
    code_lengths(fcl, 19, 7, cllen);
    canonical_codes(cllen, 19, clcode);
    static const uint8_t ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    int hclen = 19;
    while (hclen > 4 && cllen[ORDER[hclen - 1]] == 0)
        --hclen;
    {
        // characters that do not compress (random code points): the coded size would exceed the raw one -> stored blocks
        uint64_t bits = 0;
        for (int i = 0; i < 286; ++i)
            bits += (uint64_t)flit[i] * (llen[i] + (i >= 257 ? LEN_EXTRA[i - 257] : 0));
        for (int i = 0; i < 30; ++i)
            bits += (uint64_t)fdist[i] * (dlen[i] + DIST_EXTRA[i]);
        if (bits / 8 + 600 >= len) {
            out.reserve(out.size() + len + len / 65535 * 5 + 16);
            for (size_t at = 0; at < len;) {
                const size_t piece = len - at < 65535 ? len - at : 65535;
                const bool fin = last && at + piece == len;
                out.push_back(fin ? 1 : 0);  // BFINAL, BTYPE = 00, padding to the byte
                out.push_back((uint8_t)piece);
                out.push_back((uint8_t)(piece >> 8));
                out.push_back((uint8_t)~piece);
                out.push_back((uint8_t)(~piece >> 8));
                out.insert(out.end(), src + at, src + at + piece);
                at += piece;
            }
            return;
        }
    }
    out.reserve(out.size() + len / 6 + 1024);
    bit_writer bw(out);
    bw.put(0, 1);  // not the final block: the stored block below closes the chunk
    bw.put(2, 2);  // dynamic Huffman
    bw.put((uint32_t)(hlit - 257), 5);
    bw.put((uint32_t)(hdist - 1), 5);
    bw.put((uint32_t)(hclen - 4), 4);
    for (int i = 0; i < hclen; ++i)
        bw.put(cllen[ORDER[i]], 3);
    for (uint16_t c : cl) {
        const int sym = c & 0xFF;
        bw.put(clcode[sym], cllen[sym]);
        if (sym == 16)
            bw.put(c >> 8, 2);
        else if (sym == 17)
            bw.put(c >> 8, 3);
        else if (sym == 18)
            bw.put(c >> 8, 7);
    }
    for (uint32_t t : tok) {
        if (t & 0x80000000u) {
            const int length = (int)((t >> 16) & 0x1FFu), dist = (int)(t & 0xFFFFu) + 1;
            const int ls = len_symbol(length), ds = dist == 4 ? ds_unit : dist_symbol(dist);
            bw.put(lcode[257 + ls], llen[257 + ls]);
            bw.put((uint32_t)(length - LEN_BASE[ls]), LEN_EXTRA[ls]);
            bw.put(dcode[ds], dlen[ds]);
            bw.put((uint32_t)(dist - DIST_BASE[ds]), DIST_EXTRA[ds]);
        } else {
            bw.put(lcode[t], llen[t]);
        }
    }
    bw.put(lcode[256], llen[256]);
    bw.put(last ? 1 : 0, 1);  // empty stored block: final for the member's last chunk, a byte-aligning flush otherwise
    bw.put(0, 2);
    bw.align();
    out.push_back(0);
    out.push_back(0);
    out.push_back(0xFF);
    out.push_back(0xFF);
}

// Entropy coding alone (what zlib calls Z_HUFFMAN_ONLY, ~90 MB/s there): one dynamic block of literals per chunk, or
// stored blocks when even that does not pay.  For numeric members whose bytes no string match shrinks (column ids).
void encode_literals(const uint8_t *src, size_t len, bool last, std::vector<uint8_t> &out)
{
    uint32_t flit[286] = {}, fdist[30] = {};
    {
        uint32_t f4[4][256] = {};  // four histograms: consecutive bytes do not wait for each other's increment
        size_t i = 0;
        for (; i + 4 <= len; i += 4) {
            ++f4[0][src[i]];
            ++f4[1][src[i + 1]];
            ++f4[2][src[i + 2]];
            ++f4[3][src[i + 3]];
        }
        for (; i < len; ++i)
            ++f4[0][src[i]];
        for (int b = 0; b < 256; ++b)
            flit[b] = f4[0][b] + f4[1][b] + f4[2][b] + f4[3][b];
    }
    ++flit[256];
    uint8_t llen[286], dlen[30];
    uint16_t lcode[286];
    code_lengths(flit, 286, 15, llen);
    code_lengths(fdist, 30, 15, dlen);
    canonical_codes(llen, 286, lcode);
    uint64_t bits = 0;
    for (int i = 0; i < 257; ++i)
        bits += (uint64_t)flit[i] * llen[i];
    if (bits / 8 + 200 >= len) {
        out.reserve(out.size() + len + len / 65535 * 5 + 16);
        for (size_t at = 0; at < len;) {
            const size_t piece = len - at < 65535 ? len - at : 65535;
            out.push_back(last && at + piece == len ? 1 : 0);
            out.push_back((uint8_t)piece);
            out.push_back((uint8_t)(piece >> 8));
            out.push_back((uint8_t)~piece);
            out.push_back((uint8_t)(~piece >> 8));
            out.insert(out.end(), src + at, src + at + piece);
            at += piece;
        }
        return;
    }
    const int hlit = 257, hdist = 2;  // (code_lengths gave the two unused distance symbols 0, 1 one bit each)
    std::vector<uint8_t> seq(llen, llen + hlit);
    seq.insert(seq.end(), dlen, dlen + hdist);
    uint32_t fcl[19] = {};
    for (uint8_t v : seq)
        ++fcl[v];  // no run-length symbols: 259 code lengths are a few hundred bits either way
    uint8_t cllen[19];
    uint16_t clcode[19];
    code_lengths(fcl, 19, 7, cllen);
    canonical_codes(cllen, 19, clcode);
    static const uint8_t ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    int hclen = 19;
    while (hclen > 4 && cllen[ORDER[hclen - 1]] == 0)
        --hclen;
    out.reserve(out.size() + (size_t)(bits / 8) + 1024);
    bit_writer bw(out);
    bw.put(0, 1);
    bw.put(2, 2);
    bw.put((uint32_t)(hlit - 257), 5);
    bw.put((uint32_t)(hdist - 1), 5);
    bw.put((uint32_t)(hclen - 4), 4);
    for (int i = 0; i < hclen; ++i)
        bw.put(cllen[ORDER[i]], 3);
    for (uint8_t v : seq)
        bw.put(clcode[v], cllen[v]);
    for (size_t i = 0; i < len; ++i)
        bw.put(lcode[src[i]], llen[src[i]]);
    bw.put(lcode[256], llen[256]);
    bw.put(last ? 1 : 0, 1);
    bw.put(0, 2);
    bw.align();
    out.push_back(0);
    out.push_back(0);
    out.push_back(0xFF);
    out.push_back(0xFF);
}

}  // namespace utf32_deflate

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
    if (c.utf32_item != 0 && c.len >= 64) {
        utf32_deflate::encode(c.src, c.len, c.utf32_item > 0 ? c.utf32_item : 0, c.last, c.out);
        return;
    }
    // Numeric members (column ids, row starts): when a probe of the chunk's first 64 KiB says that string matching buys
    // nothing over entropy coding alone, the chunk is Huffman-coded only (5x faster, same size within 2 %)
    int strategy = Z_DEFAULT_STRATEGY;
    if (c.len >= ((size_t)1 << 17)) {
        const size_t probe = (size_t)1 << 15;
        std::vector<uint8_t> tmp(compressBound((uLong)probe) + 64);
        size_t got[2] = {0, 0};
        for (int which = 0; which < 2; ++which) {
            z_stream ps;
            memset(&ps, 0, sizeof(ps));
            if (deflateInit2(&ps, 1, Z_DEFLATED, -15, 8, which ? Z_HUFFMAN_ONLY : Z_DEFAULT_STRATEGY) != Z_OK)
                break;
            ps.next_in = const_cast<Bytef *>(c.src + c.len / 2 - probe / 2);  // the middle of the chunk
            ps.avail_in = (uInt)probe;
            ps.next_out = tmp.data();
            ps.avail_out = (uInt)tmp.size();
            if (deflate(&ps, Z_FINISH) == Z_STREAM_END)
                got[which] = ps.total_out;
            deflateEnd(&ps);
        }
        if (got[0] && got[1] && (double)got[1] <= 1.05 * (double)got[0]) {
            utf32_deflate::encode_literals(c.src, c.len, c.last, c.out);
            return;
        }
    }
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    c.err = deflateInit2(&zs, level, Z_DEFLATED, -15, 8, strategy);
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
    // 512 KiB: 16 back-reference windows (the chunk's first window starts empty: < 1 % of size), and short enough that
    // the slowest kind of chunk (zlib level 6 on near-random integers, ~17 MB/s) takes 30 ms, not 120: a 10 k-sequence
    // file has only ~200 of them to spread over the threads
    constexpr size_t CHUNK_MAX = (size_t)1 << 19, CHUNK_MIN = (size_t)1 << 17;
    unsigned hw = std::thread::hardware_concurrency();
    {
        cpu_set_t set;  // the cores this process may run on (a cgroup / taskset share of the host), not the host's
        if (sched_getaffinity(0, sizeof(set), &set) == 0 && CPU_COUNT(&set) > 0)
            hw = (unsigned)CPU_COUNT(&set);
    }
    size_t nt = nthreads >= 1 ? (size_t)nthreads : (hw ? hw : 1);
    if (nt > 64)
        nt = 64;
    // a small file (one proteome: 40 MB of arrays) has few chunks of 512 KiB in its slowest member (8 for 4 MB of column
    // ids, 4-5 ms each, while the padded strings beside them run five times faster): members are cut so that every
    // thread gets about eight chunks, down to 128 KiB (four windows).  (Measured and not kept: worker threads that
    // outlive the call and per-thread scratch buffers - the same 6-7 ms for that file, +4 % at 100 k sequences.)
    int64_t total_bytes = 0;
    for (int m = 0; m < nmembers; ++m)
        total_bytes += data_bytes && data_bytes[m] > 0 ? data_bytes[m] : 0;
    size_t CHUNK = ((size_t)(total_bytes / (int64_t)(8 * nt)) + 65535) / 65536 * 65536;
    CHUNK = CHUNK < CHUNK_MIN ? CHUNK_MIN : (CHUNK > CHUNK_MAX ? CHUNK_MAX : CHUNK);
    std::vector<npz_chunk> chunks;
    std::vector<size_t> first_chunk((size_t)nmembers + 1, 0);
    for (int m = 0; m < nmembers; ++m) {
        SKM_REQUIRE(names[m] && strlen(names[m]) > 0 && strlen(names[m]) < 60000 && header_bytes[m] > 0 && h_headers[m] &&
                        data_bytes[m] >= 0 && (data_bytes[m] == 0 || h_data[m]),
                    SKM_E_BADARG, "skm_npz_write: bad member %d", m);
        first_chunk[m] = chunks.size();
        const bool empty = data_bytes[m] == 0;
        chunks.push_back({m, (const uint8_t *)h_headers[m], (size_t)header_bytes[m], empty, {}, 0, Z_OK});
        // a little-endian UTF-32 member ('descr': '<U12'): chunks of whole items for the string encoder
        int item = 0;
        {
            const std::string hdr((const char *)h_headers[m], (size_t)header_bytes[m]);
            const size_t at = hdr.find("'descr': '<U");
            if (at != std::string::npos) {
                const long chars = strtol(hdr.c_str() + at + 12, nullptr, 10);
                if (chars > 0 && chars < ((long)1 << 28) && (int64_t)chars * 4 <= data_bytes[m] && data_bytes[m] % (chars * 4) == 0)
                    item = (int)(chars * 4);
            }
            // 4- and 8-byte little-endian integers (column ids, counts, row starts, lengths): the unit encoder finds what
            // zlib finds in them - a relative's row of column ids a few rows back, runs of equal values, elements that
            // differ in their low byte - without walking hash chains (zlib level 6: 13-25 MB/s per thread on column ids;
            // here the same size within 1-2 % at 5x the speed).  Floating-point members stay with zlib: on the sparse 0/1
            // presence matrix its chains find 15 % more (candidates whose next non-zero is as far away), at 250-350 MB/s.
            for (const char *d : {"'descr': '<u4'", "'descr': '<i4'", "'descr': '<u8'", "'descr': '<i8'"})
                if (item == 0 && hdr.find(d) != std::string::npos && data_bytes[m] % 4 == 0)
                    item = -1;
        }
        const size_t step = item > 0 && (size_t)item <= CHUNK ? CHUNK / (size_t)item * (size_t)item : CHUNK;  // (a multiple of 4)
        for (size_t at = 0; at < (size_t)data_bytes[m]; at += step) {
            const size_t len = (size_t)data_bytes[m] - at < step ? (size_t)data_bytes[m] - at : step;
            chunks.push_back({m, (const uint8_t *)h_data[m] + at, len, at + len == (size_t)data_bytes[m], {}, 0, Z_OK});
            chunks.back().utf32_item = item;
        }
    }
    first_chunk[nmembers] = chunks.size();
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

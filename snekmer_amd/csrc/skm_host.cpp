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


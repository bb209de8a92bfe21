// host_ingest.cpp -- FASTA reader, check_sequence and the 2-bit slot packer (host side; no GPU).
//
// Restates, as one pass over the file bytes, what the reference does with Python line iteration:
//   idelucs/utils.py:137-188 / :224-260  record state machine ('#' skipped, '>' flushes the current
//                                        record only if an id is set, id = line[1:-1], other lines
//                                        .strip()ped and joined, unconditional flush at EOF)
//   idelucs/utils.py:26-51               check_sequence (header checks, translate, delete, validate)
// and produces the packed slot layout documented in include/idelucs_hip.h.
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "common.h"

namespace {

struct Tables {
    uint8_t translate[256];  // utils.py:42-43 basemask; 0 = deleted (" \t\n\r"), 1 = invalid byte
    uint8_t code[256];       // kmers.pyx:19-34: 'A'0 'C'1 'G'2 'T'3, everything else 4
    Tables()
    {
        for (int i = 0; i < 256; ++i) { translate[i] = 1; code[i] = 4; }
        const char *from = "acgtuUswkmyrbdhvnSWKMYRBDHV-";
        const char *to = "ACGTTTNNNNNNNNNNNNNNNNNNNNNN";
        for (const char *c = "ACGTN"; *c; ++c) translate[(uint8_t)*c] = (uint8_t)*c;
        for (int i = 0; from[i]; ++i) translate[(uint8_t)from[i]] = (uint8_t)to[i];
        translate[(uint8_t)' '] = translate[(uint8_t)'\t'] = translate[(uint8_t)'\n'] = translate[(uint8_t)'\r'] = 0;
        code[(uint8_t)'A'] = 0; code[(uint8_t)'C'] = 1; code[(uint8_t)'G'] = 2; code[(uint8_t)'T'] = 3;
    }
};
const Tables T;

inline bool py_bytes_space(uint8_t b) { return b == ' ' || (b >= 9 && b <= 13); }  // bytes.strip() set

// chr(b) as UTF-8 (the reference formats chr(stripped[0]) into the message, utils.py:47-49)
std::string chr_utf8(uint8_t b)
{
    std::string s;
    if (b < 0x80) s.push_back((char)b);
    else { s.push_back((char)(0xC0 | (b >> 6))); s.push_back((char)(0x80 | (b & 0x3F))); }
    return s;
}

void pack_one(const uint8_t *seq, int64_t len, uint8_t *codes, uint8_t *mask)
{
    const int64_t slots = (len + 63) / 64;
    uint32_t *cw = (uint32_t *)codes;
    uint32_t *mw = (uint32_t *)mask;
    for (int64_t w = 0; w < slots * 4; ++w) {  // 16 bases per code word
        uint32_t c = 0, m = 0;
        const int64_t base = w * 16;
        for (int j = 0; j < 16; ++j) {
            const int64_t i = base + j;
            const unsigned v = (i < len) ? T.code[seq[i]] : 4u;
            if (v == 4u) m |= 0x8000u >> j;
            else c |= (uint32_t)v << (30 - 2 * j);
        }
        cw[w] = c;
        if ((w & 1) == 0) mw[w >> 1] = m << 16;
        else mw[w >> 1] |= m;
    }
}

}  // namespace

struct idl_fasta {
    std::vector<uint8_t> names;
    std::vector<int64_t> name_off{0};
    std::vector<int64_t> lengths;
    std::vector<uint8_t> bytes;
    std::vector<int64_t> byte_off{0};
    int64_t total_slots = 0;
};

extern "C" {

int idl_check_sequence(const uint8_t *in, int64_t len, uint8_t *out, int64_t *out_len, int64_t *bad_pos)
{
    IDL_REQUIRE(len >= 0 && (len == 0 || (in && out)) && out_len, "NULL buffer");
    int64_t n = 0;
    for (int64_t i = 0; i < len; ++i) {
        const uint8_t t = T.translate[in[i]];
        if (t == 0) continue;
        if (t == 1) {
            if (bad_pos) *bad_pos = i;
            idl::set_error("Invalid DNA byte: '%s'", chr_utf8(in[i]).c_str());
            return IDL_ERR_BASE;
        }
        out[n++] = t;
    }
    *out_len = n;
    return IDL_OK;
}

int idl_pack(const uint8_t *bytes, const int64_t *byte_off, int64_t n, uint8_t *codes, uint8_t *mask,
             int64_t *slot_off)
{
    IDL_REQUIRE(n >= 0 && byte_off && slot_off, "NULL buffer");
    int64_t slot = 0;
    for (int64_t s = 0; s < n; ++s) {
        const int64_t len = byte_off[s + 1] - byte_off[s];
        IDL_REQUIRE(len >= 0, "byte_off not ascending");
        slot_off[s] = slot;
        if (len > 0) {
            IDL_REQUIRE(bytes && codes && mask, "NULL buffer");
            pack_one(bytes + byte_off[s], len, codes + slot * 16, mask + slot * 8);
        }
        slot += (len + 63) / 64;
    }
    slot_off[n] = slot;
    return IDL_OK;
}

int idl_fasta_open(const char *path, int check, idl_fasta **out)
{
    IDL_REQUIRE(path && out, "NULL argument");
    *out = nullptr;
    FILE *fp = fopen(path, "rb");
    if (!fp) { idl::set_error("cannot open %s", path); return IDL_ERR_IO; }
    std::vector<uint8_t> buf;
    {
        fseek(fp, 0, SEEK_END);
        const long sz = ftell(fp);
        fseek(fp, 0, SEEK_SET);
        if (sz < 0) { fclose(fp); idl::set_error("cannot size %s", path); return IDL_ERR_IO; }
        buf.resize((size_t)sz);
        if (sz > 0 && fread(buf.data(), 1, (size_t)sz, fp) != (size_t)sz) {
            fclose(fp); idl::set_error("short read on %s", path); return IDL_ERR_IO;
        }
        fclose(fp);
    }
    idl_fasta *f = new idl_fasta();
    std::vector<uint8_t> cur;          // joined, stripped lines of the current record ("lines" in the reference)
    std::string seq_id;                // raw bytes of the id ("" = none yet)
    int rc = IDL_OK;

    auto flush = [&]() -> int {
        // utils.py:37-40 header checks (check_sequence is only called when check != 0)
        if (check) {
            if (!seq_id.empty()) {
                const uint8_t h0 = (uint8_t)seq_id[0];
                if (h0 == '>' || h0 == '#' || py_bytes_space(h0) || (h0 >= 0x1c && h0 <= 0x1f)) {
                    idl::set_error("Bad character in sequence header");
                    return IDL_ERR_HEADER;
                }
            }
            if (seq_id.find('\t') != std::string::npos) {
                idl::set_error("tab included in header");
                return IDL_ERR_HEADER;
            }
        }
        const size_t start = f->bytes.size();
        if (check) {
            f->bytes.resize(start + cur.size());
            int64_t n = 0;
            uint8_t *dst = f->bytes.data() + start;
            for (size_t i = 0; i < cur.size(); ++i) {
                const uint8_t t = T.translate[cur[i]];
                if (t == 0) continue;
                if (t == 1) {
                    idl::set_error("Invalid DNA byte in sequence %s: '%s'", seq_id.c_str(), chr_utf8(cur[i]).c_str());
                    return IDL_ERR_BASE;
                }
                dst[n++] = t;
            }
            f->bytes.resize(start + (size_t)n);
        } else {
            f->bytes.insert(f->bytes.end(), cur.begin(), cur.end());
        }
        const int64_t len = (int64_t)(f->bytes.size() - start);
        f->names.insert(f->names.end(), seq_id.begin(), seq_id.end());
        f->name_off.push_back((int64_t)f->names.size());
        f->lengths.push_back(len);
        f->byte_off.push_back((int64_t)f->bytes.size());
        f->total_slots += (len + 63) / 64;
        return IDL_OK;
    };

    const uint8_t *p = buf.data(), *end = p + buf.size();
    while (p < end) {
        const uint8_t *nl = (const uint8_t *)memchr(p, '\n', (size_t)(end - p));
        const uint8_t *le = nl ? nl + 1 : end;  // line = [p, le), includes the '\n' when present
        if (*p == '#') {
            // ignored
        } else if (*p == '>') {
            if (!seq_id.empty()) {
                rc = flush();
                if (rc != IDL_OK) break;
                cur.clear();  // NB utils.py:184/257: `lines` is reset only on a flush
            }
            // id = line[1:-1]: drops exactly one trailing byte whatever it is
            const int64_t l = le - p;
            seq_id.assign((const char *)p + 1, (size_t)(l >= 2 ? l - 2 : 0));
        } else {
            const uint8_t *a = p, *b = le;
            while (a < b && py_bytes_space(*a)) ++a;
            while (b > a && py_bytes_space(b[-1])) --b;
            cur.insert(cur.end(), a, b);
        }
        p = le;
    }
    if (rc == IDL_OK) rc = flush();  // unconditional flush at EOF (an empty file yields one empty record with id "")
    if (rc != IDL_OK) { delete f; return rc; }
    *out = f;
    return IDL_OK;
}

void idl_fasta_close(idl_fasta *f) { delete f; }

int idl_fasta_sizes(const idl_fasta *f, int64_t *n_records, int64_t *total_bases, int64_t *total_slots,
                    int64_t *names_bytes)
{
    IDL_REQUIRE(f, "NULL handle");
    if (n_records) *n_records = (int64_t)f->lengths.size();
    if (total_bases) *total_bases = (int64_t)f->bytes.size();
    if (total_slots) *total_slots = f->total_slots;
    if (names_bytes) *names_bytes = (int64_t)f->names.size();
    return IDL_OK;
}

int idl_fasta_export(const idl_fasta *f, uint8_t *names, int64_t *name_off, int64_t *lengths,
                     uint8_t *bytes, int64_t *byte_off, uint8_t *codes, uint8_t *mask, int64_t *slot_off)
{
    IDL_REQUIRE(f, "NULL handle");
    const int64_t n = (int64_t)f->lengths.size();
    if (names && !f->names.empty()) memcpy(names, f->names.data(), f->names.size());
    if (name_off) memcpy(name_off, f->name_off.data(), (size_t)(n + 1) * sizeof(int64_t));
    if (lengths && n) memcpy(lengths, f->lengths.data(), (size_t)n * sizeof(int64_t));
    if (bytes && !f->bytes.empty()) memcpy(bytes, f->bytes.data(), f->bytes.size());
    if (byte_off) memcpy(byte_off, f->byte_off.data(), (size_t)(n + 1) * sizeof(int64_t));
    if (codes) {
        IDL_REQUIRE(mask && slot_off, "codes given without mask/slot_off");
        return idl_pack(f->bytes.data(), f->byte_off.data(), n, codes, mask, slot_off);
    }
    return IDL_OK;
}

}  // extern "C"

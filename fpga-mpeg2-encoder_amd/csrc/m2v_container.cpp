// m2v_container.cpp — elementary-stream scan and MPEG-2 PS / TS multiplexers (include/m2v_container.h).
// Plain C++, no GPU: built into libm2v_container.so with g++.  Written from ISO/IEC 13818-1 (systems) and the
// start-code layer of 13818-2; the only knowledge of the encoder used here is that it writes I and P frame
// pictures in display order (RTL:2670-2682: no B pictures), so DTS = PTS and no reordering is needed.
#include "../../include/m2v_container.h"

#include <cstring>
#include <vector>

namespace {

// ---------------------------------------------------------------------------------------------
// start-code scan
// ---------------------------------------------------------------------------------------------
struct Scan {
    m2vc_stream_info info{};
    std::vector<m2vc_picture> pics;
};

// next 00 00 01 xx at or after p; returns es_bytes if none
size_t next_start_code(const uint8_t *es, size_t n, size_t p)
{
    while (p + 3 < n) {
        if (es[p + 2] > 1) { p += 3; continue; }                 // a start code cannot end here or earlier
        if (es[p] == 0 && es[p + 1] == 0 && es[p + 2] == 1) return p;
        ++p;
    }
    return n;
}

int scan(const uint8_t *es, size_t n, Scan &s)
{
    if (n < 12 || es[0] != 0 || es[1] != 0 || es[2] != 1 || es[3] != 0xB3) return M2VC_E_SYNTAX;
    m2vc_stream_info &in = s.info;
    in.width = ((uint32_t)es[4] << 4) | (es[5] >> 4);
    in.height = ((uint32_t)(es[5] & 15) << 8) | es[6];
    in.aspect_ratio_code = es[7] >> 4;
    in.frame_rate_code = es[7] & 15;
    in.bit_rate_400 = ((uint32_t)es[8] << 10) | ((uint32_t)es[9] << 2) | (es[10] >> 6);
    size_t p = 0, pending_start = (size_t)-1;                      // pending_start: GOP header waiting for its picture
    bool gop_pending = false;
    size_t end = n;
    while ((p = next_start_code(es, n, p)) < n) {
        const uint8_t code = es[p + 3];
        if (code == 0xB8) {                                        // group_of_pictures_header
            if (!s.pics.empty() && s.pics.back().bytes == 0) s.pics.back().bytes = p - s.pics.back().offset;
            in.gops++;
            gop_pending = true;
            pending_start = p;
        } else if (code == 0x00) {                                 // picture_header
            if (p + 6 > n) return M2VC_E_SYNTAX;
            if (!s.pics.empty() && s.pics.back().bytes == 0) s.pics.back().bytes = p - s.pics.back().offset;
            m2vc_picture pic{};
            pic.offset = gop_pending ? pending_start : p;
            pic.gop_start = gop_pending ? 1u : 0u;
            pic.temporal_reference = ((uint32_t)es[p + 4] << 2) | (es[p + 5] >> 6);
            pic.coding_type = (es[p + 5] >> 3) & 7u;
            if (pic.coding_type == 1) in.i_pictures++;
            else if (pic.coding_type == 2) in.p_pictures++;
            else return M2VC_E_SYNTAX;                             // the encoder writes I and P pictures only
            s.pics.push_back(pic);
            in.pictures++;
            gop_pending = false;
        } else if (code >= 0x01 && code <= 0xAF) {                 // slice
            if (s.pics.empty()) return M2VC_E_SYNTAX;
            s.pics.back().slices++;
            in.slices++;
        } else if (code == 0xB7) {                                 // sequence_end_code
            if (!s.pics.empty() && s.pics.back().bytes == 0) s.pics.back().bytes = p - s.pics.back().offset;
            in.has_sequence_end = 1;
            end = p + 4;
            break;
        }
        p += 4;
    }
    if (!s.pics.empty() && s.pics.back().bytes == 0) s.pics.back().bytes = end - s.pics.back().offset;
    in.bytes = end;
    for (size_t i = end; i < n; ++i)
        if (es[i] != 0) return M2VC_E_SYNTAX;                      // only zero padding may follow (RTL:2932-2937)
    in.padding_bytes = n - end;
    return M2VC_OK;
}

const uint32_t kRateNum[9] = {0, 24000, 24, 25, 30000, 30, 50, 60000, 60};
const uint32_t kRateDen[9] = {1, 1001, 1, 1, 1001, 1, 1, 1001, 1};

// 90 kHz ticks per picture as a rational: 90000 * den / num
struct Clock {
    uint64_t num, den;
    uint64_t pts(uint64_t picture, uint64_t base) const { return base + picture * 90000ull * den / num; }
};

// ---------------------------------------------------------------------------------------------
// bit writer for the fixed-layout headers
// ---------------------------------------------------------------------------------------------
struct Bits {
    std::vector<uint8_t> &v;
    uint64_t acc = 0;
    int n = 0;
    explicit Bits(std::vector<uint8_t> &out) : v(out) {}
    void put(uint64_t val, int len)
    {
        for (int i = len - 1; i >= 0; --i) {
            acc = (acc << 1) | ((val >> i) & 1u);
            if (++n == 8) { v.push_back((uint8_t)acc); acc = 0; n = 0; }
        }
    }
};

void put_timestamp(Bits &b, uint32_t prefix4, uint64_t t)         // '0010' / '0011' / '0001' + 33 bits + 3 markers
{
    b.put(prefix4, 4);
    b.put((t >> 30) & 7u, 3);  b.put(1, 1);
    b.put((t >> 15) & 0x7FFFu, 15);  b.put(1, 1);
    b.put(t & 0x7FFFu, 15);  b.put(1, 1);
}

int finish(const std::vector<uint8_t> &v, uint8_t *out, size_t cap, size_t *out_bytes)
{
    if (out_bytes) *out_bytes = v.size();
    if (!out) return M2VC_OK;
    if (cap < v.size()) return M2VC_E_OVERFLOW;
    memcpy(out, v.data(), v.size());
    return M2VC_OK;
}

// ---------------------------------------------------------------------------------------------
// program stream
// ---------------------------------------------------------------------------------------------
constexpr size_t kPackBytes = 2048;

void ps_pack_header(std::vector<uint8_t> &v, uint64_t scr27, uint32_t mux_rate50)
{
    Bits b(v);
    const uint64_t base = (scr27 / 300) & 0x1FFFFFFFFull, ext = scr27 % 300;
    b.put(0x000001BAu, 32);
    b.put(1, 2);
    b.put((base >> 30) & 7u, 3);  b.put(1, 1);
    b.put((base >> 15) & 0x7FFFu, 15);  b.put(1, 1);
    b.put(base & 0x7FFFu, 15);  b.put(1, 1);
    b.put(ext, 9);  b.put(1, 1);
    b.put(mux_rate50, 22);  b.put(3, 2);
    b.put(0x1F, 5);  b.put(0, 3);                                  // reserved, pack_stuffing_length = 0
}

void ps_system_header(std::vector<uint8_t> &v, uint32_t mux_rate50, uint32_t vbuf_kb)
{
    Bits b(v);
    b.put(0x000001BBu, 32);
    b.put(9, 16);                                                  // header_length
    b.put(1, 1);  b.put(mux_rate50, 22);  b.put(1, 1);             // rate_bound
    b.put(0, 6);                                                   // audio_bound
    b.put(0, 1);  b.put(0, 1);                                     // fixed_flag, CSPS_flag
    b.put(0, 1);  b.put(1, 1);  b.put(1, 1);                       // audio lock, video lock, marker
    b.put(1, 5);                                                   // video_bound
    b.put(0, 1);  b.put(0x7F, 7);                                  // packet_rate_restriction_flag, reserved
    b.put(0xE0, 8);  b.put(3, 2);  b.put(1, 1);  b.put(vbuf_kb, 13);   // P-STD buffer bound, scale 1 = 1024 bytes
}

}  // namespace

extern "C" {

int m2vc_frame_rate(uint32_t code, uint32_t *num, uint32_t *den)
{
    if (code == 0 || code > 8) { if (num) *num = 0; if (den) *den = 1; return M2VC_E_PARAM; }
    if (num) *num = kRateNum[code];
    if (den) *den = kRateDen[code];
    return M2VC_OK;
}

int m2vc_scan(const uint8_t *es, size_t es_bytes, m2vc_stream_info *info, m2vc_picture *pics, size_t cap, size_t *npics)
{
    if (!es || !info) return M2VC_E_PARAM;
    Scan s;
    const int r = scan(es, es_bytes, s);
    *info = s.info;
    if (r < 0) return r;
    if (npics) *npics = s.pics.size();
    if (pics) {
        const size_t k = s.pics.size() < cap ? s.pics.size() : cap;
        memcpy(pics, s.pics.data(), k * sizeof(m2vc_picture));
        if (k < s.pics.size()) return M2VC_E_OVERFLOW;
    }
    return M2VC_OK;
}

int m2vc_mux_ps(const uint8_t *es, size_t es_bytes, uint8_t *out, size_t cap, size_t *out_bytes)
{
    if (!es) return M2VC_E_PARAM;
    Scan s;
    int r = scan(es, es_bytes, s);
    if (r < 0) return r;
    uint32_t fn, fd;
    if (m2vc_frame_rate(s.info.frame_rate_code, &fn, &fd) < 0 || s.pics.empty()) return M2VC_E_SYNTAX;
    const Clock clk{fn, fd};
    const size_t n = (size_t)s.info.bytes;                         // the zero padding after sequence_end_code is dropped

    // multiplex rate: the whole stream in the time its pictures take to display, +10 % for the pack overhead,
    // at least 1 Mbit/s; in units of 50 bytes/s
    const double seconds = (double)s.pics.size() * fd / fn;
    double bytes_per_s = (double)n / seconds * 1.10;
    if (bytes_per_s < 125000.0) bytes_per_s = 125000.0;
    const uint32_t mux_rate50 = (uint32_t)((bytes_per_s + 49.0) / 50.0);
    const double rate = mux_rate50 * 50.0;
    uint64_t maxpic = 0;
    for (auto &p : s.pics) maxpic = p.bytes > maxpic ? p.bytes : maxpic;
    uint32_t vbuf_kb = (uint32_t)((2 * maxpic + 1023) / 1024 + 16);
    if (vbuf_kb > 8191) vbuf_kb = 8191;
    // presentation starts once the largest picture can have arrived twice over, plus one picture period
    const uint64_t pts0 = (uint64_t)(2.0 * (double)maxpic / rate * 90000.0) + clk.pts(1, 0) + 900;

    std::vector<uint8_t> v;
    v.reserve(n + n / 64 + 4096);
    size_t pos = 0, next_pic = 0;
    while (pos < n) {
        const size_t pack_start = v.size();
        ps_pack_header(v, (uint64_t)((double)pack_start / rate * 27000000.0), mux_rate50);
        if (pack_start == 0) ps_system_header(v, mux_rate50, vbuf_kb);
        // every picture starts a PES packet of its own (the sequence headers travel with the first one), so every
        // picture has a PTS and the payload of a packet with a PTS starts with a start code (data_alignment_indicator);
        // packs are 2048 bytes except where a picture ends earlier
        auto start_of = [&](size_t i) -> size_t { return i == 0 ? 0 : (size_t)s.pics[i].offset; };
        while (next_pic < s.pics.size() && start_of(next_pic) < pos) ++next_pic;
        const bool has_pts = next_pic < s.pics.size() && start_of(next_pic) == pos;
        const size_t following = has_pts ? next_pic + 1 : next_pic;
        const size_t limit = following < s.pics.size() ? start_of(following) : n;
        size_t payload = kPackBytes - (v.size() - pack_start) - 9 - (has_pts ? 5 : 0);
        if (payload > limit - pos) payload = limit - pos;
        Bits b(v);
        b.put(0x000001E0u, 32);
        b.put(3 + (has_pts ? 5 : 0) + payload, 16);                // PES_packet_length
        b.put(2, 2);  b.put(0, 2);  b.put(0, 1);                   // '10', scrambling, priority
        b.put(has_pts ? 1 : 0, 1);                                 // data_alignment_indicator
        b.put(0, 1);  b.put(1, 1);                                 // copyright, original
        b.put(has_pts ? 2 : 0, 2);  b.put(0, 6);                   // PTS_DTS_flags, no other optional fields
        b.put(has_pts ? 5 : 0, 8);                                 // PES_header_data_length
        if (has_pts) put_timestamp(b, 2, clk.pts(next_pic, pts0));
        v.insert(v.end(), es + pos, es + pos + payload);
        pos += payload;
    }
    const uint8_t endc[4] = {0, 0, 1, 0xB9};                       // MPEG_program_end_code
    v.insert(v.end(), endc, endc + 4);
    return finish(v, out, cap, out_bytes);
}

// ---------------------------------------------------------------------------------------------
// transport stream
// ---------------------------------------------------------------------------------------------
static uint32_t crc32_mpeg(const uint8_t *p, size_t n)
{
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; ++i) {
        c ^= (uint32_t)p[i] << 24;
        for (int k = 0; k < 8; ++k) c = (c & 0x80000000u) ? (c << 1) ^ 0x04C11DB7u : (c << 1);
    }
    return c;
}

namespace {
constexpr uint32_t kPidPmt = 0x1000, kPidVideo = 0x100;

struct TsWriter {
    std::vector<uint8_t> &v;
    uint8_t cc_pat = 0, cc_pmt = 0, cc_vid = 0;
    explicit TsWriter(std::vector<uint8_t> &out) : v(out) {}

    void psi(uint32_t pid, uint8_t &cc, const std::vector<uint8_t> &section)
    {
        const size_t at = v.size();
        v.resize(at + 188, 0xFF);
        uint8_t *p = v.data() + at;
        p[0] = 0x47;
        p[1] = 0x40 | (uint8_t)(pid >> 8);                         // payload_unit_start_indicator
        p[2] = (uint8_t)pid;
        p[3] = 0x10 | (cc++ & 15);                                 // payload only
        p[4] = 0;                                                  // pointer_field
        memcpy(p + 5, section.data(), section.size());
    }

    void pat()
    {
        std::vector<uint8_t> s = {0x00, 0xB0, 13, 0x00, 0x01, 0xC1, 0x00, 0x00,         // table 0, length, ts id 1, version 0/current
                                  0x00, 0x01, (uint8_t)(0xE0 | (kPidPmt >> 8)), (uint8_t)kPidPmt};
        const uint32_t c = crc32_mpeg(s.data(), s.size());
        for (int k = 3; k >= 0; --k) s.push_back((uint8_t)(c >> (8 * k)));
        psi(0, cc_pat, s);
    }

    void pmt()
    {
        std::vector<uint8_t> s = {0x02, 0xB0, 18, 0x00, 0x01, 0xC1, 0x00, 0x00,         // table 2, program 1
                                  (uint8_t)(0xE0 | (kPidVideo >> 8)), (uint8_t)kPidVideo, 0xF0, 0x00,   // PCR PID, no program info
                                  0x02, (uint8_t)(0xE0 | (kPidVideo >> 8)), (uint8_t)kPidVideo, 0xF0, 0x00};  // MPEG-2 video
        const uint32_t c = crc32_mpeg(s.data(), s.size());
        for (int k = 3; k >= 0; --k) s.push_back((uint8_t)(c >> (8 * k)));
        psi(kPidPmt, cc_pmt, s);
    }

    // one PES packet (header + payload) as video TS packets; the first one carries the PCR
    void pes(const std::vector<uint8_t> &hdr, const uint8_t *payload, size_t n, uint64_t pcr27)
    {
        size_t hpos = 0, ppos = 0;
        bool first = true;
        while (hpos < hdr.size() || ppos < n) {
            const size_t left = (hdr.size() - hpos) + (n - ppos);
            const size_t at = v.size();
            v.resize(at + 188, 0xFF);
            uint8_t *p = v.data() + at;
            p[0] = 0x47;
            p[1] = (first ? 0x40 : 0x00) | (uint8_t)(kPidVideo >> 8);
            p[2] = (uint8_t)kPidVideo;
            size_t af = first ? 8 : 0;                             // adaptation field bytes incl. the length byte
            if (184 - af > left) af = 184 - left;                  // stuffing so the payload ends with the packet
            if (af == 1) { p[4] = 0; }                             // adaptation_field_length = 0
            else if (af >= 2) {
                p[4] = (uint8_t)(af - 1);
                p[5] = first ? 0x10 : 0x00;                        // PCR_flag
                size_t q = 6;
                if (first) {
                    const uint64_t base = (pcr27 / 300) & 0x1FFFFFFFFull, ext = pcr27 % 300;
                    p[6] = (uint8_t)(base >> 25);  p[7] = (uint8_t)(base >> 17);  p[8] = (uint8_t)(base >> 9);
                    p[9] = (uint8_t)(base >> 1);   p[10] = (uint8_t)(((base & 1) << 7) | 0x7E | (ext >> 8));
                    p[11] = (uint8_t)ext;
                    q = 12;
                }
                (void)q;                                           // the rest stays 0xFF stuffing
            }
            p[3] = (uint8_t)((af ? 0x30 : 0x10) | (cc_vid++ & 15));
            size_t w = 4 + af;
            while (w < 188 && hpos < hdr.size()) p[w++] = hdr[hpos++];
            const size_t take = 188 - w < n - ppos ? 188 - w : n - ppos;
            memcpy(p + w, payload + ppos, take);
            ppos += take;
            first = false;
        }
    }
};
}  // namespace

int m2vc_mux_ts(const uint8_t *es, size_t es_bytes, uint8_t *out, size_t cap, size_t *out_bytes)
{
    if (!es) return M2VC_E_PARAM;
    Scan s;
    int r = scan(es, es_bytes, s);
    if (r < 0) return r;
    uint32_t fn, fd;
    if (m2vc_frame_rate(s.info.frame_rate_code, &fn, &fd) < 0 || s.pics.empty()) return M2VC_E_SYNTAX;
    const Clock clk{fn, fd};
    const size_t n = (size_t)s.info.bytes;
    uint64_t maxpic = 0;
    for (auto &p : s.pics) maxpic = p.bytes > maxpic ? p.bytes : maxpic;
    const double seconds = (double)s.pics.size() * fd / fn;
    double rate = (double)n / seconds * 1.15;                      // bytes/s of the transport stream
    if (rate < 125000.0) rate = 125000.0;
    const uint64_t pts0 = (uint64_t)(2.0 * (double)maxpic / rate * 90000.0) + clk.pts(1, 0) + 900;

    std::vector<uint8_t> v;
    v.reserve(n + n / 16 + 4096);
    TsWriter w(v);
    double next_psi = 0.0;
    for (size_t i = 0; i < s.pics.size(); ++i) {
        const double now = (double)v.size() / rate;
        if (now >= next_psi) { w.pat(); w.pmt(); next_psi = now + 0.1; }
        // the first picture's PES packet also carries the sequence headers in front of it
        const size_t a = i == 0 ? 0 : (size_t)s.pics[i].offset;
        const size_t b = i + 1 < s.pics.size() ? (size_t)s.pics[i + 1].offset : n;
        std::vector<uint8_t> hdr;
        Bits hb(hdr);
        hb.put(0x000001E0u, 32);
        hb.put(0, 16);                                             // PES_packet_length 0: unbounded, video in a transport stream
        hb.put(2, 2);  hb.put(0, 2);  hb.put(0, 1);  hb.put(1, 1);  hb.put(0, 1);  hb.put(1, 1);
        hb.put(2, 2);  hb.put(0, 6);
        hb.put(5, 8);
        put_timestamp(hb, 2, clk.pts(i, pts0));
        w.pes(hdr, es + a, b - a, (uint64_t)((double)v.size() / rate * 27000000.0));
    }
    return finish(v, out, cap, out_bytes);
}

}  // extern "C"

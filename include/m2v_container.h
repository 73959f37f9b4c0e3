/*
 * m2v_container.h — stream-level conveniences around the encoder's output (libm2v_container.so, plain C++,
 * no GPU): a structural scan of the elementary stream (sizes / types per picture: the numbers the reference's
 * README:735-768 quotes per clip) and MPEG-2 Program Stream / Transport Stream multiplexers so that the `.m2v`
 * the module produces (SIM/tb_mpeg2encoder.v:260-262 writes the bare elementary stream) can be played and muxed
 * with audio by ordinary tools.  SURVEY.md 8(f4): usefulness beyond parity; nothing here touches the encoded bits.
 *
 * All functions return 0 / a count on success and a negative M2VC_E_* code on failure.  Buffers are caller owned.
 */
#ifndef M2V_CONTAINER_H
#define M2V_CONTAINER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    M2VC_OK         = 0,
    M2VC_E_PARAM    = -1,
    M2VC_E_SYNTAX   = -2,   /* not an MPEG-2 video elementary stream as this encoder writes it */
    M2VC_E_OVERFLOW = -3    /* output buffer / picture table too small */
};

typedef struct {
    uint32_t width, height;          /* sequence_header horizontal/vertical_size_value (RTL:2600-2603) */
    uint32_t frame_rate_code;        /* ISO/IEC 13818-2 table 6-4; the RTL writes 2 = 24 fps (RTL:2598-2617) */
    uint32_t aspect_ratio_code;
    uint32_t bit_rate_400;           /* bit_rate_value, units of 400 bit/s (0x3FFFF = variable) */
    uint32_t pictures, i_pictures, p_pictures, gops, slices;
    uint64_t bytes;                  /* stream length without the zero padding after sequence_end_code */
    uint64_t padding_bytes;          /* zero bytes after sequence_end_code (RTL:2932-2937 pads to 32-byte words) */
    int      has_sequence_end;
} m2vc_stream_info;

typedef struct {
    uint64_t offset;                 /* byte offset of the picture's first start code (GOP header if one precedes it) */
    uint64_t bytes;                  /* up to the next picture / sequence end */
    uint32_t coding_type;            /* 1 = I, 2 = P */
    uint32_t temporal_reference;
    uint32_t gop_start;              /* 1 if a group_of_pictures_header precedes the picture */
    uint32_t slices;
} m2vc_picture;

/* frames per second of a frame_rate_code as a rational (0/1 for reserved codes) */
int m2vc_frame_rate(uint32_t frame_rate_code, uint32_t *num, uint32_t *den);

/*
 * Structural scan.  `pics` may be NULL (count only); otherwise up to `cap` entries are filled and *npics receives
 * the number of pictures in the stream (M2VC_E_OVERFLOW if cap was too small, the first cap entries are valid).
 */
int m2vc_scan(const uint8_t *es, size_t es_bytes, m2vc_stream_info *info, m2vc_picture *pics, size_t cap,
              size_t *npics);

/*
 * Both multiplexers are for FILE PLAYBACK: SCR / PCR advance with the byte position at a constant multiplex rate a little
 * above the stream's average rate and a picture's PTS is its index times the picture period; the delivery is not paced
 * against the P-STD / T-STD buffer model (a very large first picture can arrive after its PTS, a long stream runs ahead of
 * the decoder).  Software players and remultiplexers accept that; a hardware or strictly STD-checking demultiplexer needs
 * the stream re-paced by a real multiplexer.
 *
 * MPEG-2 Program Stream (ISO/IEC 13818-1 2.5): packs of at most 2048 bytes, a system header in the first pack, one
 * video PES stream (stream_id 0xE0); every picture starts a PES packet that carries its PTS (no B pictures: DTS =
 * PTS; the sequence headers travel with the first picture), MPEG_program_end_code at the end.  The multiplex
 * rate is derived from the stream size and the frame rate.
 * Call with out = NULL to get the required size in *out_bytes.
 */
int m2vc_mux_ps(const uint8_t *es, size_t es_bytes, uint8_t *out, size_t cap, size_t *out_bytes);

/*
 * MPEG-2 Transport Stream (ISO/IEC 13818-1 2.4): 188-byte packets, PAT (PID 0) + PMT (PID 0x1000) repeated every
 * 0.1 s of stream time, video on PID 0x100 carrying the PCR, one PES packet per picture with PTS.
 * Call with out = NULL to get the required size in *out_bytes.
 */
int m2vc_mux_ts(const uint8_t *es, size_t es_bytes, uint8_t *out, size_t cap, size_t *out_bytes);

#ifdef __cplusplus
}
#endif
#endif

/* agc_read.h -- C ABI of the read side, library libagc_read.so.
 *
 * Same entry points, argument meaning and return values as the C section of the reference's
 * src/lib-cxx/agc-api.h:118-213 (implementation src/lib-cxx/lib-cxx.cpp:123-330), so a program written
 * against the reference's libagc links against this library unchanged.  Host code only: decoding an archive
 * (zstd + LZ-diff decode + reverse complement + stitching) is not part of the GPU hot path.
 * Two additions, marked "extension", return FASTA text the way `agc getset` / `agc getctg` write it.
 */
#ifndef AGC_READ_H
#define AGC_READ_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct agc_t agc_t;

/* agc-api.h:118  NULL for error; prefetching is accepted and ignored (the file is always mapped whole) */
agc_t *agc_open(char *fn, int prefetching);
/* agc-api.h:125  0 for success, -1 for error; frees the handle */
int agc_close(agc_t *agc);
/* agc-api.h:138  contig length or < 0 (-1 unknown, -2 name not unique and sample NULL) */
int agc_get_ctg_len(const agc_t *agc, const char *sample, const char *name);
/* agc-api.h:150  [start, end] inclusive; buf must hold end - start + 2 bytes; returns the length written or < 0 */
int agc_get_ctg_seq(const agc_t *agc, const char *sample, const char *name, int start, int end, char *buf);
/* agc-api.h:157 */
int agc_n_sample(const agc_t *agc);
/* agc-api.h:165 */
int agc_n_ctg(const agc_t *agc, const char *sample);
/* agc-api.h:172  malloc'ed string, free with agc_string_destroy */
char *agc_reference_sample(const agc_t *agc);
/* agc-api.h:180  NULL-terminated malloc'ed array (sorted sample names), free with agc_list_destroy */
char **agc_list_sample(const agc_t *agc, int *n_sample);
/* agc-api.h:189 */
char **agc_list_ctg(const agc_t *agc, const char *sample, int *n_ctg);
/* agc-api.h:196 */
int agc_list_destroy(char **list);
/* agc-api.h:203 */
int agc_string_destroy(char *sample);

/* extension: whole sample as FASTA text (`agc getset`), malloc'ed, *len = bytes; NULL for error */
char *agc_get_sample_fasta(const agc_t *agc, const char *sample, int line_length, long long *len);
/* extension: k, min_match_len, pack_cardinality, segment_size stored in the archive; 0 for success */
int agc_get_params(const agc_t *agc, unsigned *k, unsigned *min_match_len, unsigned *pack_cardinality, unsigned *segment_size);

#ifdef __cplusplus
}
#endif
#endif

/*
 * oracle/loftr_oracle.h -- TEST INFRASTRUCTURE ONLY (see loftr_oracle.c).
 * CPU restatement of ::DNNFeatureMatcher::MatchFrames (/root/reference/src/dnnfeaturematcher.cpp:44-102)
 * for the fixed-shape 1x1x480x640 LoFTR_teacher graph.
 */
#ifndef ORACLE_LOFTR_ORACLE_H
#define ORACLE_LOFTR_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct loftr_oracle loftr_oracle;
int loftr_oracle_set_threads(int n);   /* returns the thread count in effect */
loftr_oracle* loftr_oracle_create(const char* weights_blob_path);
void loftr_oracle_destroy(loftr_oracle* o);
/* conf: [1200*1200] (required); sim [1200*1200], feat0/feat1 [1200*32], tok [2*1200*32] optional (NULL) */
int loftr_oracle_run(loftr_oracle* o, const uint8_t* im0, ptrdiff_t s0, const uint8_t* im1, ptrdiff_t s1,
                     float* conf, float* sim, float* feat0, float* feat1, float* tok);
int loftr_oracle_decode(const float* conf, float threshold, int32_t* out, int cap);
int loftr_oracle_match(loftr_oracle* o, const uint8_t* im0, ptrdiff_t s0, const uint8_t* im1, ptrdiff_t s1,
                       float threshold, int32_t* out, int cap);
#ifdef __cplusplus
}
#endif
#endif

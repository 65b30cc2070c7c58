/* oracle/frame.c -- TEST INFRASTRUCTURE (see jxl_oracle.h).
 * Whole-frame driver: DecodeGroup for every group, then the stage list of
 * PassesDecoderState::PreparePipeline (lib/jxl/dec_cache.cc:151-170,259-265):
 * [Gaborish] [EPF0 if iters>=3] [EPF1 if >=1] [EPF2 if >=2] XYB, executed
 * stage-at-a-time like SimpleRenderPipeline::ProcessBuffers
 * (simple_render_pipeline.cc:79-297).  pthreads over groups / rows stand in
 * for the JxlParallelRunner (lib/jxl/dec_frame.cc:694-731). */
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "jxl_oracle.h"

typedef struct {
  const jxo_frame* f;
  int kind; /* 0 groups, 1 gab, 2..4 epf0..2, 5 xyb->rgb, 6 copy xyb out */
  float* const* a;
  float* const* b;
  size_t stride;
  const float* sigma;
  float* out;
  size_t out_stride, out_plane_stride;
  uint32_t begin, end;
  int rc;
} job;

static void* run_job(void* arg) {
  job* j = (job*)arg;
  const float* const* in = (const float* const*)j->a;
  switch (j->kind) {
    case 0: j->rc = jxo_decode_groups(j->f, j->a, j->stride, j->begin, j->end); break;
    case 1: jxo_gaborish(j->f, in, j->b, j->stride, j->begin, j->end); break;
    case 2: case 3: case 4:
      jxo_epf(j->f, j->kind - 2, j->sigma, in, j->b, j->stride, j->begin, j->end);
      break;
    case 5:
      jxo_xyb_to_linear_rgb(j->f, in, j->stride, j->out, j->out_stride, j->begin, j->end);
      break;
    case 6:
      for (int c = 0; c < 3; c++)
        for (uint32_t y = j->begin; y < j->end; y++)
          memcpy(j->out + c * j->out_plane_stride + (size_t)y * j->out_stride,
                 j->a[c] + (size_t)y * j->stride, sizeof(float) * j->f->p.xsize);
      break;
  }
  return NULL;
}

static int parallel(job proto, uint32_t total, int threads) {
  if (threads < 1) threads = 1;
  if ((uint32_t)threads > total) threads = total ? (int)total : 1;
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * threads);
  job* jobs = (job*)malloc(sizeof(job) * threads);
  for (int t = 0; t < threads; t++) {
    jobs[t] = proto;
    jobs[t].begin = (uint32_t)((uint64_t)total * t / threads);
    jobs[t].end = (uint32_t)((uint64_t)total * (t + 1) / threads);
    jobs[t].rc = 0;
    if (threads == 1) run_job(&jobs[t]);
    else pthread_create(&th[t], NULL, run_job, &jobs[t]);
  }
  int rc = 0;
  for (int t = 0; t < threads; t++) {
    if (threads > 1) pthread_join(th[t], NULL);
    if (jobs[t].rc) rc = jobs[t].rc;
  }
  free(th);
  free(jobs);
  return rc;
}

int jxo_decode_frame(const jxo_frame* f, float* out, size_t out_stride,
                     size_t out_plane_stride, int threads) {
  const jxlhip_frame_params* p = &f->p;
  const uint32_t xsb = (p->xsize + 7) / 8, ysb = (p->ysize + 7) / 8;
  const uint32_t ngroups = ((p->xsize + 255) / 256) * ((p->ysize + 255) / 256);
  const size_t stride = (size_t)xsb * 8, rows = (size_t)ysb * 8;
  float* A[3];
  float* B[3];
  for (int c = 0; c < 3; c++) {
    A[c] = (float*)calloc(stride * rows, sizeof(float));
    B[c] = (float*)calloc(stride * rows, sizeof(float));
    if (!A[c] || !B[c]) return -1;
  }
  float* sigma = NULL;
  job j;
  memset(&j, 0, sizeof(j));
  j.f = f;
  j.stride = stride;
  j.kind = 0;
  j.a = A;
  int rc = parallel(j, ngroups, threads);
  float** cur = A;
  float** nxt = B;
  if (rc == 0) {
    if (p->lf.epf_iters > 0) {
      sigma = (float*)calloc((size_t)xsb * ysb, sizeof(float));
      jxo_compute_sigma(f, sigma);
    }
    int stages[4], ns = 0;
    if (p->lf.gab) stages[ns++] = 1;
    if (p->lf.epf_iters >= 3) stages[ns++] = 2;
    if (p->lf.epf_iters >= 1) stages[ns++] = 3;
    if (p->lf.epf_iters >= 2) stages[ns++] = 4;
    for (int s = 0; s < ns; s++) {
      j.kind = stages[s];
      j.a = cur;
      j.b = nxt;
      j.sigma = sigma;
      parallel(j, p->ysize, threads);
      float** t = cur;
      cur = nxt;
      nxt = t;
    }
    j.a = cur;
    j.out = out;
    j.out_stride = out_stride;
    j.out_plane_stride = out_plane_stride;
    j.kind = p->output_kind == JXLHIP_OUT_XYB_PLANAR ? 6 : 5;
    float* lin = NULL;
    if (p->output_kind == JXLHIP_OUT_PACKED) {
      /* linear RGB rows first, then FromLinear + WriteToOutput (output.c);
       * out_stride is in BYTES for this kind */
      lin = (float*)calloc((size_t)p->xsize * p->ysize * 3, sizeof(float));
      if (!lin) rc = -1;
      j.out = lin;
      j.out_stride = (size_t)p->xsize * 3;
    }
    if (rc == 0) parallel(j, p->ysize, threads);
    if (lin) {
      jxo_pack_output(f, lin, (size_t)p->xsize * 3, out, out_stride, 0, p->ysize);
      free(lin);
    }
  }
  for (int c = 0; c < 3; c++) {
    free(A[c]);
    free(B[c]);
  }
  free(sigma);
  return rc;
}

// dua_denoiser_step: one denoiser evaluation + sampler update behind a single C entry point (include/dua_hip.h).
//
// Host-side sequencing only: every launch goes through the public entry points of this library, in the order the
// caller's plan lists them.  This is the loop body the reference runs per reverse-diffusion step
// (guided_diffusion/gaussian_diffusion.py:487-535 / 667-716 around models/basic_unet/denoiser.py:284-312), so a
// maintainer who binds the shared library gets "the step", not 47 kernels to order; Python captures exactly this call
// into the HIP graph it replays.
#include <hip/hip_runtime.h>
#include "../../include/dua_hip.h"

extern "C" int dua_denoiser_step(const dua_denoiser_plan* p, void* stream) {
  if (!p || p->N <= 0 || !p->ops || p->n_ops <= 0 || !p->stat_arena || p->stat_bytes <= 0 || !p->tail_raw) return DUA_ERR_ARG;
  // step begin and the clear of the statistics arena are ONE launch (see step_begin_kernel: a memset node is not)
  int rc = dua_step_begin_clear(p->N, p->P, p->temb_table, p->table_rows, p->rows_per_sample, p->row_of_step, p->nsteps,
                                p->coef_table, p->counter, p->cur_add, p->cur_coef, p->step_word, p->err_word,
                                p->stat_arena, p->stat_bytes, stream);
  if (rc) return rc;
  for (int i = 0; i < p->n_ops; ++i) {
    const dua_step_op& o = p->ops[i];
    const dua_in_norm* in = o.has_norm ? &o.norm : nullptr;
    switch (o.kind) {
      case DUA_OP_CONV3:
        rc = dua_conv3d_k3_fwd(&o.conv, o.x, o.w, o.bias, in, o.y, o.stats, p->workspace, p->workspace_bytes, stream);
        break;
      case DUA_OP_MATERIALIZE:
        if (!in) return DUA_ERR_ARG;
        rc = dua_materialize(&o.mat, o.x, in, o.emb, o.y, o.pooled, stream);
        break;
      case DUA_OP_DECONV:
        rc = dua_deconv_k2s2_fwd(&o.conv, o.x, o.w, o.bias, in, o.y, stream);
        break;
      case DUA_OP_UPCONV:
        rc = dua_upconv_k3_fwd(&o.up, o.x, o.u, in, o.w, o.wu, o.bias, o.y, o.stats, stream);
        break;
      default:
        return DUA_ERR_ARG;
    }
    if (rc) return rc;
  }
  return dua_final_conv_sampler(&p->tail, p->tail_raw, &p->tail_norm, p->wf, p->bf, p->cur_coef, p->x_state, p->noise,
                                p->step_word, p->xin, p->xstart_sum, p->logits, p->xstart, stream);
}

// libgemmgan_lab.so: registers the opt-in kernels of this library (ffn.hip, enc.hip, head.hip) with the engine in libgemmgan.so when the
// library is loaded (include/gemmgan_lab.h).  The kernel-level test hooks of testhooks.hip are built into the same library.
#include "../../include/gemmgan_lab.h"
#include "gg_common.h"
#include "kernels.h"

namespace {
int g_loaded = 0;
__attribute__((constructor)) void register_lab() {
    gg::LabTable t;
    t.ffn_fused_supported = &gg::ffn_fused_supported;
    t.ffn_fused = &gg::ffn_fused;
    t.ffn2_supported = &gg::ffn2_supported;
    t.ffn2 = &gg::ffn2;
    t.ffn2_sweep_tokens = &gg::ffn2_sweep_tokens;
    t.k_enc_frag_weights = &gg::k_enc_frag_weights;
    t.enc_bwd_supported = &gg::enc_bwd_supported;
    t.enc_bwd = &gg::enc_bwd;
    t.k_encb_frag_weights = &gg::k_encb_frag_weights;
    t.head_fused_supported = &gg::head_fused_supported;
    t.head_fwd = &gg::head_fwd;
    t.head_bwd = &gg::head_bwd;
    g_loaded = gg_lab_register(&t, sizeof t) == 0;
}
}  // namespace

extern "C" int gg_lab_loaded(void) { return g_loaded; }

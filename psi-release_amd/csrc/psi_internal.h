// Host-side launchers shared between translation units of libpsi_hip.so (not part of the C ABI).
#pragma once
#include "psi_common.h"

struct psi_lbs_model;

// Optional per-stage HIP-event timer (psi_fit_profile): when active, every kernel launch site marks an event on the
// launch stream right after the launch; inactive (nullptr) in normal operation and under graph capture.
struct PsiStageTimer {
    hipEvent_t ev[48];
    const char *name[48];
    int n;
};
extern thread_local PsiStageTimer *g_psi_timer;
static inline void psi_mark(const char *nm, hipStream_t st)
{
    PsiStageTimer *t = g_psi_timer;
    if (t && t->n < 48) {
        (void)hipEventRecord(t->ev[t->n], st);
        t->name[t->n] = nm;
        t->n++;
    }
}

// chamfer.hip
int psi_nn_contact(const float *verts, long vstride, const int *vid, const float *scene, int B, int n, int m, void *ws,
                   float cconst, float gscale, float *gq, float *fpart, int *idx_out, hipStream_t st);
int psi_nn_contact_fparts(int n);
size_t psi_nn_ws_bytes(int B, int n, int m);

// nnindex.hip
struct psi_nn_index;
int psi_nn_index_contact(const psi_nn_index *ix, const float *verts, long vstride, const int *vid, int B, int n, float cconst,
                         float gscale, float *gq, float *fpart, int *hint, hipStream_t st);
int psi_nn_index_fparts(int n);

// lbs.hip
struct PsiLbsGradOut {
    float *g_betas;    // [B,NB]
    float *g_pose;     // [B,J*3]  axis-angle gradient (through Rodrigues)
    float *g_transl;   // [B,3]
    float *g_rot;      // [B,J,9]  gradient wrt the rotation matrices themselves (before the Rodrigues derivative), nullable
};
int psi_lbs_backward_ex(const psi_lbs_model *mdl, const float *grad_verts, const float *betas, const float *pose,
                        const float *cam_ext, int B, float *ws, PsiLbsGradOut out, hipStream_t st);
void psi_lbs_dims(const psi_lbs_model *mdl, int *V, int *J, int *NB);
// the fused fitting engine runs the per-body pose stages inside its own kernels (lbs_device.h) and calls these for the rest
struct PsiLbsView;
int psi_lbs_view(const psi_lbs_model *mdl, int B, float *ws, PsiLbsView *out);
int psi_lbs_blend_forward(const psi_lbs_model *mdl, int B, float *ws, hipStream_t st);          // v_posed = v_t + feat @ dirs
int psi_lbs_backward_joint_parts(const psi_lbs_model *mdl, int B, float *ws, float *g_transl, hipStream_t st,
                                 bool reduce = true);   // skin_bwd_A + blend_bwd partials (+ their reduction: gA, gfeat in the workspace)

// dp.hip
struct psi_dp_comm;
int psi_dp_world(const psi_dp_comm *c);

// conv_stem.hip: the 2 -> 64 channel 7x7 stride-2 stem convolution (forward, weight gradient); conv_gemm.hip's entry points route to it
bool psi_conv_stem_shape(int Cin, int Cout, int kh, int kw, int stride, int pad);
int psi_conv_stem_forward(const void *x, int x_bf16, const float *w, const float *bias, int N, int H, int W, void *y, int y_bf16, int nterm,
                          hipStream_t st);
size_t psi_conv_stem_wgrad_floats(int N, int H, int W);
int psi_conv_stem_weight_grad(const void *x, int x_bf16, const void *dy, int dy_bf16, int N, int H, int W, float *gw, float *ws, int nterm,
                              hipStream_t st);

// conv.hip: the stride-1 3x3 convolutions of the fp32 model (three-term split products) with the input tile split ONCE per workgroup;
// conv_gemm.hip's entry points route the shapes these cover here
int psi_conv3x3_wrw3_ok(int N, int H, int W, int Cin, int Cout);
size_t psi_conv3x3_wrw3_workspace_floats(int N, int H, int W, int Cin, int Cout);
int psi_conv3x3_weight_grad3(const float *x, const float *dy, int N, int H, int W, int Cin, int Cout, float *gw, float *ws, hipStream_t st);
int psi_conv3x3s_ok(int N, int H, int W, int Cin, int Cout);
// wp: prepared weight parts (psi_conv2d_prepare_weight) — forward layout [Cout][9][Cin]; reversed_taps = 1 with the input gradient's layout
// [Cin][9][Cout] (and Cin / Cout, x / y in the roles of dY's channels / dX): the input gradient of the same layer
int psi_conv3x3s_forward(const float *x, const void *wp, const float *bias, int N, int H, int W, int Cin, int Cout, float *y, int reversed_taps, hipStream_t st);

// Host-side executor: the plan that replaces SqueezeDet._add_forward_graph /
// SqueezeDetPlus._add_forward_graph / ResNet50ConvDet._add_forward_graph (reference
// src/nets/squeezeDet.py:30-79, src/nets/squeezeDetPlus.py:30-79, src/nets/resnet50_convDet.py:31-169).
// It owns no device memory: packed parameters and the activation workspace are bound by the
// caller (sqdet_net_bind).
#include <math.h>
#include <string.h>

#include <string>
#include <vector>

#include "common.h"
#include "chain.h"

namespace sqdet {
int conv2d_launch(const void* x, const void* w_packed, const float* bias, void* y, int n, int h, int w, int cin,
                  int cout, int k, int stride, int pad_mode, int relu, int dtype, int y_cstride, int y_coffset,
                  hipStream_t st);
int conv2d_launch_ex(const void* x, const void* w_packed, const float* bias, void* y, int n, int h, int w, int cin,
                     int cout, int k, int stride, int pad_mode, int relu, int dtype, int y_cstride, int y_coffset,
                     int x_cstride, int x_coffset, int accum, hipStream_t st);
int convdet_scored_launch(const void* x, const void* w_packed, const float* bias, void* preds, float* scores, int n, int h, int w,
                          int cin, int apg, int classes, int dtype, hipStream_t st);
int fold_bn_launch(const float* w, const float* cbias, const float* gamma, const float* beta, const float* mean,
                   const float* var, float eps, float* wf, float* bf, int k, int cin, int cout, hipStream_t st);
int maxpool_launch(const void* x, void* y, int n, int h, int w, int c, int k, int stride, int pad_mode, int dtype,
                   hipStream_t st);
int stem_launch(const void* x, const void* w_packed, const float* bias, void* y, int n, int h, int w, int cout, int k,
                int conv_pad, int pool_pad, int dtype, int y_cstride, int y_coffset, hipStream_t st, bool* handled);
int stem_squeeze_launch(const void* x, const void* w_packed, const float* bias, const void* ws2_packed, const float* bs2,
                        void* s_out, int n, int h, int w, int cout, int k, int conv_pad, int pool_pad, int s2, int dtype,
                        hipStream_t st, bool* handled);
bool stem_squeeze_eligible(int h, int w, int cout, int k, int conv_pad, int pool_pad, int s2, int dtype, int n);
int fire_fused_launch_keep(const void* x, const void* ws, const float* bs, const void* w1, const float* b1, const void* w3,
                           const float* b3, void* sq_out, void* y, int n, int h, int w, int cin, int s, int e1, int e3, int dtype,
                           hipStream_t st, bool* handled);
int fire_fused_launch(const void* x, const void* ws, const float* bs, const void* w1, const float* b1, const void* w3,
                      const float* b3, void* y, int n, int h, int w, int cin, int s, int e1, int e3, int dtype,
                      hipStream_t st, bool* handled);
bool fire_fused_eligible(int cin, int s, int e1, int e3, int dtype);
bool fire_stream_eligible(int cin, int s, int e1, int e3, int dtype);
int fire_stream_launch_ex(const void* x, const void* ws, const float* bs, const void* w1, const float* b1, const void* w3,
                          const float* b3, void* y, int n, int h, int w, int cin, int s, int e1, int e3, int dtype,
                          int pool, hipStream_t st, bool* handled);
bool fire_chain_eligible(int s, int e1, int e3, int s2, int dtype);
bool fire_expand_stream_eligible(int s, int e1, int e3, int dtype);
bool fire_squeeze_next_eligible(int cin, int s, int e1, int e3, int s2, int dtype);
bool fire_expand_squeeze_next_eligible(int s, int e1, int e3, int s2, int pool, int dtype);
int fire_expand_squeeze_next_launch(const void* sq_in, const void* w1, const float* b1, const void* w3, const float* b3,
                                    const void* ws2, const float* bs2, void* s_out, int n, int h, int w, int s, int e1, int e3,
                                    int s2, int pool, int dtype, hipStream_t st, bool* handled);
int fire_squeeze_next_launch(const void* x, const void* ws, const float* bs, const void* w1, const float* b1, const void* w3,
                             const float* b3, const void* ws2, const float* bs2, void* s_out, int n, int h, int w, int cin,
                             int s, int e1, int e3, int s2, int dtype, hipStream_t st, bool* handled);
int fire_expand_stream_launch(const void* sq_in, const void* w1, const float* b1, const void* w3, const float* b3, void* y,
                              int n, int h, int w, int s, int e1, int e3, int dtype, int pool, hipStream_t st, bool* handled);
bool conv3x3_pair_eligible(int n, int h, int w, int s, int e1, int e3, int dtype);
int conv3x3_pair_launch(const void* sq_in, const void* w3, const float* b3, const void* w1, const float* b1, void* y, int n, int h, int w,
                        int s, int e1, int e3, int dtype, hipStream_t st, bool* handled);
int conv_algo();
int tune(int which);
}  // namespace sqdet

using namespace sqdet;

namespace {

enum { BUF_INPUT = -1, BUF_PREDS = -2, BUF_A = 0, BUF_B = 1, BUF_S = 2, BUF_T = 3, NUM_BUFS = 4 };
// L_STEM: conv(k, s2, Cin 3)+relu+maxpool(3, s2) in one launch; L_FIRE: squeeze + both expands in one launch;
// L_CHAIN: both expands of a fire module + the squeeze of the NEXT module in one launch (sqdet_fire_chain_fwd)
// L_EXPAND: both expands of a fire module from its squeeze tensor (+ the max-pool behind it): sqdet_fire_expand_fwd
// L_FIRESQ: a whole fire module from x whose output is the NEXT module's squeeze tensor (sqdet_fire_squeeze_next_fwd)
// L_EXPSQ: both expands (+ pool) of a module from its squeeze tensor, output = the NEXT module's squeeze tensor
// L_STEMSQ: conv1 + pool1 + the first module's squeeze1x1 in the persistent stem launch (sqdet_stem_conv_pool_squeeze_fwd)
enum { L_CONV = 0, L_POOL = 1, L_STEM = 2, L_FIRE = 3, L_CHAIN = 4, L_EXPAND = 5, L_FIRESQ = 6, L_EXPSQ = 7, L_STEMSQ = 8 };

struct Param {
  std::string name;
  int shape[4];
  int ndim;
  size_t offset;  // into param_mem
  size_t bytes;
  int fold = -1;  // index into sqdet_net::folds when the parameter belongs to a _conv_bn_layer
};

// One _conv_bn_layer (nn_skeleton.py:374-468): the float32 kernel and the BN vectors are kept as
// set; the packed kernel + folded bias the conv launch reads are rebuilt when any of them changed.
struct BnFold {
  int k, cin, cout;
  int kparam, cbias, gamma, beta, mean, var;  // params indices (cbias -1: conv_with_bias=False)
  size_t raw_off;    // float32 HWIO copy of the kernel
  size_t fbias_off;  // folded bias, float32 [cout]
  bool dirty;
};

struct Layer {
  int type;
  std::string name;
  int in_buf, out_buf;
  int h, w, cin, cout, k, stride, pad_mode, relu;
  int ho, wo;
  int y_cstride, y_coffset;
  int kparam, bparam;  // indices into params (conv)
  int fold = -1;       // BnFold index: bias comes from the folded-bias slot
  int accum = 0;       // y = act(conv + bias + y): the residual add of a bottleneck block
  int pool_pad_mode;   // L_STEM: padding of the fused pool
  // L_FIRE: cin = fire input channels, cout = e1 + e3; the three convs' parameters
  int fs, fe1, fe3;
  int kp_s, bp_s, kp_1, bp_1, kp_3, bp_3;
  int fire_pool = 0;   // L_FIRE: the 3x3/s2 SAME max-pool that follows is taken inside the kernel (ho, wo = pooled dims)
  // L_CHAIN: in_buf holds the module's squeeze tensor (fs channels); fs2 > 0: the next module's squeeze tensor goes to
  // out_buf, else the concat tensor does; kp_s2 / bp_s2 = the next module's squeeze parameters; the packed stream
  int fs2 = 0, kp_s2 = -1, bp_s2 = -1;
  size_t chain_off = 0;
  double flops, bytes;
};

struct FireSpec { const char* name; int s, e1, e3; };

}  // namespace

struct sqdet_net {
  int arch, dtype, batch, img_h, img_w, classes, apg;
  std::vector<Param> params;
  std::vector<Layer> layers;
  size_t param_bytes = 0;
  std::vector<BnFold> folds;
  size_t fold_scratch_off = 0, fold_scratch_bytes = 0;  // folded float32 kernel before packing
  float bn_eps = 1e-5f;                                 // config/config.py:131
  size_t buf_elems[NUM_BUFS] = {0, 0, 0, 0};
  size_t buf_off[NUM_BUFS] = {0, 0, 0, 0};
  size_t workspace_bytes = 0;
  int gh = 0, gw = 0, out_ch = 0;
  char* param_mem = nullptr;
  char* workspace = nullptr;
  std::vector<hipEvent_t> events;
  // live probe: start/stop events around ONE layer's launch inside sqdet_net_forward
  int probe_layer = -1;
  int probe_count = 0;
  std::vector<hipEvent_t> probe_events;  // 2 per record
  // sqdet_net_set_scores: when not NULL the ConvDet launch (last layer) also writes interpret_output's det_probs there
  float* scores = nullptr;
  // sqdet_net_set_post_job: the previous batch's decode + filter, carried by the NEXT forward's fire_chain launches as rider
  // workgroups (chain.h); one-shot
  bool job_set = false;
  sqdet::FilterArgs job_fa;
  sqdet::DecodeArgs job_da;
  int job_n = 0;
  // sqdet_net_set_signal: an event recorded right before layer signal_layer's launch
  int signal_layer = -1;
  hipEvent_t signal_event = nullptr;
};

namespace {

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Builder {
  sqdet_net* net;
  int h, w, c;      // current activation dims
  int cur;          // current buffer id
  size_t esz;

  int add_param(const std::string& name, int ndim, const int* shape, size_t bytes) {
    Param p;
    p.name = name;
    p.ndim = ndim;
    for (int i = 0; i < 4; ++i) p.shape[i] = i < ndim ? shape[i] : 1;
    p.offset = net->param_bytes;
    p.bytes = bytes;
    net->param_bytes = align_up(net->param_bytes + bytes, 256);
    net->params.push_back(p);
    return (int)net->params.size() - 1;
  }

  void note_buf(int buf, size_t elems) {
    if (buf >= 0 && elems > net->buf_elems[buf]) net->buf_elems[buf] = elems;
  }

  // conv reading `in_buf` (dims h,w,cin) writing channels [coff, coff+cout) of out_buf rows of cstride channels
  void conv(const std::string& name, int in_buf, int out_buf, int cin, int cout, int k, int stride, int pad_mode,
            int relu, int cstride, int coff) {
    Layer L;
    L.type = L_CONV;
    L.name = name;
    L.in_buf = in_buf; L.out_buf = out_buf;
    L.h = h; L.w = w; L.cin = cin; L.cout = cout; L.k = k; L.stride = stride; L.pad_mode = pad_mode; L.relu = relu;
    L.ho = out_size(h, k, stride, pad_mode);
    L.wo = out_size(w, k, stride, pad_mode);
    L.y_cstride = cstride; L.y_coffset = coff;
    const int kshape[4] = {k, k, cin, cout};
    L.kparam = add_param(name + "/kernels", 4, kshape, sqdet_conv_packed_bytes(k, cin, cout, net->dtype));
    const int bshape[1] = {cout};
    L.bparam = add_param(name + "/biases", 1, bshape, (size_t)cout * 4);
    const double npix = (double)net->batch * L.ho * L.wo;
    L.flops = 2.0 * k * k * cin * cout * npix;
    L.bytes = ((double)net->batch * h * w * cin + npix * cout + (double)k * k * cin * cout) * (double)esz + cout * 4.0;
    note_buf(out_buf, (size_t)net->batch * L.ho * L.wo * cstride);
    net->layers.push_back(L);
  }

  // _conv_bn_layer (nn_skeleton.py:374-468): variables kernels, [biases], gamma, beta, mean, var in
  // the reference's creation order; executed as one conv with the BN folded in.
  void conv_bn(const std::string& name, int in_buf, int out_buf, int cin, int cout, int k, int stride, int relu,
               bool with_bias, int accum) {
    Layer L;
    L.type = L_CONV;
    L.name = name;
    L.in_buf = in_buf; L.out_buf = out_buf;
    L.h = h; L.w = w; L.cin = cin; L.cout = cout; L.k = k; L.stride = stride; L.pad_mode = SQDET_PAD_SAME; L.relu = relu;
    L.ho = out_size(h, k, stride, SQDET_PAD_SAME);
    L.wo = out_size(w, k, stride, SQDET_PAD_SAME);
    L.y_cstride = cout; L.y_coffset = 0;
    L.accum = accum;
    BnFold f;
    f.k = k; f.cin = cin; f.cout = cout; f.dirty = true;
    const int kshape[4] = {k, k, cin, cout};
    const int vshape[1] = {cout};
    const size_t raw_bytes = (size_t)k * k * cin * cout * 4;
    f.kparam = L.kparam = add_param(name + "/kernels", 4, kshape, sqdet_conv_packed_bytes(k, cin, cout, net->dtype));
    f.cbias = with_bias ? add_param(name + "/biases", 1, vshape, (size_t)cout * 4) : -1;
    f.gamma = add_param(name + "/gamma", 1, vshape, (size_t)cout * 4);
    f.beta = add_param(name + "/beta", 1, vshape, (size_t)cout * 4);
    f.mean = add_param(name + "/mean", 1, vshape, (size_t)cout * 4);
    f.var = add_param(name + "/var", 1, vshape, (size_t)cout * 4);
    f.raw_off = net->param_bytes;
    net->param_bytes = align_up(net->param_bytes + raw_bytes, 256);
    f.fbias_off = net->param_bytes;
    net->param_bytes = align_up(net->param_bytes + (size_t)cout * 4, 256);
    if (raw_bytes > net->fold_scratch_bytes) net->fold_scratch_bytes = raw_bytes;
    L.bparam = -1;
    L.fold = (int)net->folds.size();
    for (int pi : {f.kparam, f.cbias, f.gamma, f.beta, f.mean, f.var})
      if (pi >= 0) net->params[pi].fold = L.fold;
    net->folds.push_back(f);
    const double npix = (double)net->batch * L.ho * L.wo;
    L.flops = 2.0 * k * k * cin * cout * npix;
    L.bytes = ((double)net->batch * h * w * cin + npix * cout * (accum ? 2.0 : 1.0) + (double)k * k * cin * cout) *
                  (double)esz + cout * 4.0;
    note_buf(out_buf, (size_t)net->batch * L.ho * L.wo * cout);
    net->layers.push_back(L);
  }

  // One bottleneck of ResNet50ConvDet (resnet50_convDet.py:50-118 + _res_branch :134-169):
  //   out = relu(shortcut + branch2c(branch2b(branch2a(x)))), shortcut = branch1(x) in the 'a'
  // blocks, x itself otherwise; the stride sits on branch1 / branch2a (caffe-style).  branch2c
  // adds into the shortcut buffer in its epilogue, so identity blocks update x in place.
  void res_block(const std::string& stage, const std::string& n, int in_f, int out_f, bool down, bool has_branch1) {
    const std::string blk = stage + "/res" + n + "/";
    const std::string b2 = blk + "res" + n + "_branch2/res" + n;
    const int X = cur, Y = other(cur);
    const int stride = down ? 2 : 1;
    const int h0 = h, w0 = w;
    if (has_branch1) conv_bn(blk + "res" + n + "_branch1", X, Y, c, out_f, 1, stride, 0, false, 0);
    conv_bn(b2 + "_branch2a", X, BUF_S, c, in_f, 1, stride, 1, false, 0);
    h = out_size(h0, 1, stride, SQDET_PAD_SAME); w = out_size(w0, 1, stride, SQDET_PAD_SAME);
    conv_bn(b2 + "_branch2b", BUF_S, BUF_T, in_f, in_f, 3, 1, 1, false, 0);
    conv_bn(b2 + "_branch2c", BUF_T, has_branch1 ? Y : X, in_f, out_f, 1, 1, 1, false, 1);
    c = out_f;
    if (has_branch1) cur = Y;
  }

  int other(int buf) { return buf == BUF_A ? BUF_B : BUF_A; }

  void conv_layer(const std::string& name, int cout, int k, int stride, int pad_mode, int relu, bool last) {
    const int out = last ? BUF_PREDS : (cur == BUF_INPUT ? BUF_A : other(cur));
    conv(name, cur, out, c, cout, k, stride, pad_mode, relu, cout, 0);
    const Layer& L = net->layers.back();
    h = L.ho; w = L.wo; c = cout; cur = out;
  }

  void pool_layer(const std::string& name, int k, int stride, int pad_mode) {
    Layer L;
    L.type = L_POOL;
    L.name = name;
    L.in_buf = cur; L.out_buf = other(cur);
    L.h = h; L.w = w; L.cin = c; L.cout = c; L.k = k; L.stride = stride; L.pad_mode = pad_mode; L.relu = 0;
    L.ho = out_size(h, k, stride, pad_mode);
    L.wo = out_size(w, k, stride, pad_mode);
    L.y_cstride = c; L.y_coffset = 0; L.kparam = L.bparam = -1;
    L.flops = 0;
    L.bytes = ((double)net->batch * h * w * c + (double)net->batch * L.ho * L.wo * c) * (double)esz;
    note_buf(L.out_buf, (size_t)net->batch * L.ho * L.wo * c);
    net->layers.push_back(L);
    h = L.ho; w = L.wo; cur = L.out_buf;
  }

  // SqueezeDet._fire_layer (nets/squeezeDet.py:81-106): squeeze -> S; expand1x1 / expand3x3
  // write the two halves of the concat tensor directly (no concat pass).
  void fire_layer(const FireSpec& f) {
    const std::string n = f.name;
    const int out = other(cur);
    conv(n + "/squeeze1x1", cur, BUF_S, c, f.s, 1, 1, SQDET_PAD_SAME, 1, f.s, 0);
    conv(n + "/expand1x1", BUF_S, out, f.s, f.e1, 1, 1, SQDET_PAD_SAME, 1, f.e1 + f.e3, 0);
    conv(n + "/expand3x3", BUF_S, out, f.s, f.e3, 3, 1, SQDET_PAD_SAME, 1, f.e1 + f.e3, f.e1);
    c = f.e1 + f.e3;
    cur = out;
  }
};

const FireSpec kSqueezeDetFires[] = {{"fire2", 16, 64, 64},    {"fire3", 16, 64, 64},    {"fire4", 32, 128, 128},
                                     {"fire5", 32, 128, 128},  {"fire6", 48, 192, 192},  {"fire7", 48, 192, 192},
                                     {"fire8", 64, 256, 256},  {"fire9", 64, 256, 256},  {"fire10", 96, 384, 384},
                                     {"fire11", 96, 384, 384}};
const FireSpec kSqueezeDetPlusFires[] = {{"fire2", 96, 64, 64},     {"fire3", 96, 64, 64},     {"fire4", 192, 128, 128},
                                         {"fire5", 192, 128, 128},  {"fire6", 288, 192, 192},  {"fire7", 288, 192, 192},
                                         {"fire8", 384, 256, 256},  {"fire9", 384, 256, 256},  {"fire10", 384, 256, 256},
                                         {"fire11", 384, 256, 256}};

void* buf_ptr(const sqdet_net* net, int buf, const void* input, void* preds) {
  if (buf == BUF_INPUT) return const_cast<void*>(input);
  if (buf == BUF_PREDS) return preds;
  return net->workspace + net->buf_off[buf];
}

// Runs layer L on images [n0, n0 + nb) of the batch (activations are NHWC with the image index outermost, so a
// sub-batch is a pointer offset).
// riders one fire_chain launch takes: its idle CUs ("dbg" 300 + k caps it at k -- experiments)
int layer_riders(const sqdet_net* net, const Layer& L) {
  int idle = sqdet::fire_chain_idle_cus(net->batch, L.h, L.w, L.fs, L.fe1, L.fe3, L.fs2, net->dtype);
  const int d = sqdet::tune(5);
  if (d >= 300 && d < 400 && idle > d - 300) idle = d - 300;
  return idle;
}

int run_layer_part(sqdet_net* net, const Layer& L, const void* input, void* preds, int n0, int nb, hipStream_t st) {
  const size_t esz = dtype_size(net->dtype);
  if (L.type == L_CHAIN) {
    const size_t px0 = (size_t)n0 * L.h * L.w;
    const char* sq_in = reinterpret_cast<const char*>(buf_ptr(net, L.in_buf, input, preds)) + px0 * L.fs * esz;
    char* out = reinterpret_cast<char*>(buf_ptr(net, L.out_buf, input, preds)) + px0 * (L.fs2 > 0 ? L.fs2 : L.fe1 + L.fe3) * esz;
    auto pb = [&](int i) { return i >= 0 ? reinterpret_cast<const float*>(net->param_mem + net->params[i].offset) : nullptr; };
    // riders: the pending job's images are dealt over the plan's fire_chain launches from the LAST one backwards (the late
    // launches are the long ones), one image per idle CU and launch
    sqdet::ChainRide ride;
    const sqdet::ChainRide* rp = nullptr;
    if (net->job_set && n0 == 0 && nb == net->batch) {
      const int idle = layer_riders(net, L);
      int later = 0;      // images taken by the chain launches behind this one
      for (size_t i = net->layers.size(); i-- > 0 && &net->layers[i] != &L;)
        if (net->layers[i].type == L_CHAIN) later += layer_riders(net, net->layers[i]);
      const int left = net->job_n - later;
      if (idle > 0 && left > 0) {
        ride.fa = net->job_fa; ride.da = net->job_da;
        ride.nimg = left < idle ? left : idle;
        ride.img0 = left - ride.nimg;
        ride.nriders = ride.nimg;
        rp = &ride;
      }
    }
    return sqdet::fire_chain_launch_ride(sq_in, net->param_mem + L.chain_off, pb(L.bp_1), pb(L.bp_3), pb(L.bp_s2),
                                         L.fs2 > 0 ? nullptr : out, L.fs2 > 0 ? out : nullptr, nb, L.h, L.w, L.fs, L.fe1, L.fe3, L.fs2,
                                         net->dtype, rp, st);
  }
  if (L.type == L_FIRESQ) {
    const char* xin = reinterpret_cast<const char*>(buf_ptr(net, L.in_buf, input, preds)) + (size_t)n0 * L.h * L.w * L.cin * esz;
    char* out = reinterpret_cast<char*>(buf_ptr(net, L.out_buf, input, preds)) + (size_t)n0 * L.h * L.w * L.fs2 * esz;
    auto pk = [&](int i) { return (const void*)(net->param_mem + net->params[i].offset); };
    auto pb = [&](int i) { return reinterpret_cast<const float*>(net->param_mem + net->params[i].offset); };
    return sqdet_fire_squeeze_next_fwd(xin, pk(L.kp_s), pb(L.bp_s), pk(L.kp_1), pb(L.bp_1), pk(L.kp_3), pb(L.bp_3), pk(L.kp_s2),
                                       pb(L.bp_s2), out, nb, L.h, L.w, L.cin, L.fs, L.fe1, L.fe3, L.fs2, net->dtype,
                                       reinterpret_cast<sqdet_stream_t>(st));
  }
  if (L.type == L_EXPSQ) {
    const char* sq_in = reinterpret_cast<const char*>(buf_ptr(net, L.in_buf, input, preds)) + (size_t)n0 * L.h * L.w * L.fs * esz;
    char* out = reinterpret_cast<char*>(buf_ptr(net, L.out_buf, input, preds)) + (size_t)n0 * L.ho * L.wo * L.fs2 * esz;
    auto pk = [&](int i) { return (const void*)(net->param_mem + net->params[i].offset); };
    auto pb = [&](int i) { return reinterpret_cast<const float*>(net->param_mem + net->params[i].offset); };
    return sqdet_fire_expand_squeeze_next_fwd(sq_in, pk(L.kp_1), pb(L.bp_1), pk(L.kp_3), pb(L.bp_3), pk(L.kp_s2), pb(L.bp_s2), out, nb,
                                              L.h, L.w, L.fs, L.fe1, L.fe3, L.fs2, L.fire_pool, net->dtype,
                                              reinterpret_cast<sqdet_stream_t>(st));
  }
  if (L.type == L_EXPAND) {
    const char* sq_in = reinterpret_cast<const char*>(buf_ptr(net, L.in_buf, input, preds)) + (size_t)n0 * L.h * L.w * L.fs * esz;
    char* out = reinterpret_cast<char*>(buf_ptr(net, L.out_buf, input, preds)) + (size_t)n0 * L.ho * L.wo * (L.fe1 + L.fe3) * esz;
    auto pk = [&](int i) { return (const void*)(net->param_mem + net->params[i].offset); };
    auto pb = [&](int i) { return reinterpret_cast<const float*>(net->param_mem + net->params[i].offset); };
    return sqdet_fire_expand_fwd(sq_in, pk(L.kp_1), pb(L.bp_1), pk(L.kp_3), pb(L.bp_3), out, nb, L.h, L.w, L.fs, L.fe1, L.fe3,
                                 L.fire_pool, net->dtype, reinterpret_cast<sqdet_stream_t>(st));
  }
  if (L.type == L_STEMSQ) {
    const void* x = reinterpret_cast<const char*>(buf_ptr(net, L.in_buf, input, preds)) + (size_t)n0 * L.h * L.w * 3 * esz;
    void* so = reinterpret_cast<char*>(buf_ptr(net, L.out_buf, input, preds)) + (size_t)n0 * L.ho * L.wo * L.fs2 * esz;
    const void* wp = net->param_mem + net->params[L.kparam].offset;
    const float* b = reinterpret_cast<const float*>(net->param_mem + net->params[L.bparam].offset);
    const void* ws = net->param_mem + net->params[L.kp_s2].offset;
    const float* bs = reinterpret_cast<const float*>(net->param_mem + net->params[L.bp_s2].offset);
    bool handled = false;
    const int rc = stem_squeeze_launch(x, wp, b, ws, bs, so, nb, L.h, L.w, L.cout, L.k, L.pad_mode, L.pool_pad_mode, L.fs2,
                                       net->dtype, st, &handled);
    if (rc != SQDET_OK) return rc;
    if (!handled) { set_error("net: stem + squeeze launch no longer eligible (options changed after net_create?)"); return SQDET_ESTATE; }
    return SQDET_OK;
  }
  const int in_c = L.type == L_STEM ? 3 : L.cin;
  const void* x = reinterpret_cast<const char*>(buf_ptr(net, L.in_buf, input, preds)) + (size_t)n0 * L.h * L.w * in_c * esz;
  void* y = reinterpret_cast<char*>(buf_ptr(net, L.out_buf, input, preds)) +
            (size_t)n0 * L.ho * L.wo * (L.type == L_POOL ? L.cin : L.y_cstride) * esz;
  if (L.type == L_CONV) {
    const void* wp = net->param_mem + net->params[L.kparam].offset;
    const float* b = reinterpret_cast<const float*>(
        net->param_mem + (L.fold >= 0 ? net->folds[L.fold].fbias_off : net->params[L.bparam].offset));
    if (net->scores && &L == &net->layers.back())   // (validated by sqdet_net_set_scores)
      return convdet_scored_launch(x, wp, b, y, net->scores + (size_t)n0 * L.h * L.w * net->apg, nb, L.h, L.w, L.cin, net->apg,
                                   net->classes, net->dtype, st);
    return conv2d_launch_ex(x, wp, b, y, nb, L.h, L.w, L.cin, L.cout, L.k, L.stride, L.pad_mode, L.relu,
                            net->dtype, L.y_cstride, L.y_coffset, L.cin, 0, L.accum, st);
  }
  if (L.type == L_STEM) {
    const void* wp = net->param_mem + net->params[L.kparam].offset;
    const float* b = reinterpret_cast<const float*>(
        net->param_mem + (L.fold >= 0 ? net->folds[L.fold].fbias_off : net->params[L.bparam].offset));
    bool handled = false;
    const int rc = stem_launch(x, wp, b, y, nb, L.h, L.w, L.cout, L.k, L.pad_mode, L.pool_pad_mode, net->dtype,
                               L.y_cstride, L.y_coffset, st, &handled);
    if (rc != SQDET_OK) return rc;
    if (!handled) { set_error("net: fused stem no longer eligible (conv_algo changed after net_create?)"); return SQDET_ESTATE; }
    return SQDET_OK;
  }
  if (L.type == L_FIRE) {
    auto pk = [&](int i) { return (const void*)(net->param_mem + net->params[i].offset); };
    auto pb = [&](int i) { return reinterpret_cast<const float*>(net->param_mem + net->params[i].offset); };
    bool handled = false;
    const int rc = L.fire_pool
        ? fire_stream_launch_ex(x, pk(L.kp_s), pb(L.bp_s), pk(L.kp_1), pb(L.bp_1), pk(L.kp_3), pb(L.bp_3), y, nb, L.h, L.w,
                                L.cin, L.fs, L.fe1, L.fe3, net->dtype, 1, st, &handled)
        : fire_fused_launch(x, pk(L.kp_s), pb(L.bp_s), pk(L.kp_1), pb(L.bp_1), pk(L.kp_3), pb(L.bp_3), y,
                            nb, L.h, L.w, L.cin, L.fs, L.fe1, L.fe3, net->dtype, st, &handled);
    if (rc != SQDET_OK) return rc;
    if (!handled) { set_error("net: fused fire no longer eligible (options changed after net_create?)"); return SQDET_ESTATE; }
    return SQDET_OK;
  }
  return maxpool_launch(x, y, nb, L.h, L.w, L.cin, L.k, L.stride, L.pad_mode, net->dtype, st);
}

int run_layer(sqdet_net* net, const Layer& L, const void* input, void* preds, hipStream_t st) {
  return run_layer_part(net, L, input, preds, 0, net->batch, st);
}

// Re-folds (sqdet_fold_batchnorm) and re-packs every _conv_bn_layer whose parameters changed since
// the last forward; stream-ordered before the layers that read them.
int refresh_folds(sqdet_net* net, hipStream_t st) {
  for (BnFold& f : net->folds) {
    if (!f.dirty) continue;
    auto vec = [&](int pi) { return pi >= 0 ? reinterpret_cast<const float*>(net->param_mem + net->params[pi].offset) : nullptr; };
    float* scratch = reinterpret_cast<float*>(net->param_mem + net->fold_scratch_off);
    int rc = fold_bn_launch(reinterpret_cast<const float*>(net->param_mem + f.raw_off), vec(f.cbias), vec(f.gamma),
                            vec(f.beta), vec(f.mean), vec(f.var), net->bn_eps, scratch,
                            reinterpret_cast<float*>(net->param_mem + f.fbias_off), f.k, f.cin, f.cout, st);
    if (rc != SQDET_OK) return rc;
    rc = sqdet_conv_pack_weights(scratch, net->param_mem + net->params[f.kparam].offset, f.k, f.cin, f.cout, net->dtype,
                                 reinterpret_cast<sqdet_stream_t>(st));
    if (rc != SQDET_OK) return rc;
    f.dirty = false;
  }
  return SQDET_OK;
}

// squeeze1x1 + expand1x1 + expand3x3 -> one L_FIRE launch (decided at plan creation) wherever a fused kernel takes the
// module.  (Rounds 1-4 fused maps of <= 100000 pixels only -- tuned on SqueezeDet, whose large maps run the streaming
// kernels anyway; SqueezeDet+'s 227 k-pixel fire2-4 at batch 8 run 1.5-2 % faster fused, SqueezeDet is level:
// profiles/r05_fire_fuse_ab.txt.)  "fire_fuse" = 2 disables the fusion, 10 restores the pixel rule.
void fuse_fires(sqdet_net* net, size_t esz) {
  if (conv_algo() != 0 || tune(3) == 2) return;
  std::vector<Layer> out;
  const std::vector<Layer>& in = net->layers;
  for (size_t i = 0; i < in.size(); ++i) {
    const bool trio = i + 2 < in.size() && in[i].type == L_CONV && in[i + 1].type == L_CONV && in[i + 2].type == L_CONV &&
                      in[i].out_buf == BUF_S && in[i + 1].in_buf == BUF_S && in[i + 2].in_buf == BUF_S && in[i].k == 1 &&
                      in[i + 1].k == 1 && in[i + 2].k == 3 && in[i + 1].out_buf == in[i + 2].out_buf;
    const long pixels = (long)net->batch * in[i].h * in[i].w;
    const bool streams = trio && tune(3) != 3 && fire_stream_eligible(in[i].cin, in[i].cout, in[i + 1].cout, in[i + 2].cout, net->dtype);
    if (!trio || !(tune(3) != 10 || pixels <= 100000 || streams) ||
        !fire_fused_eligible(in[i].cin, in[i].cout, in[i + 1].cout, in[i + 2].cout, net->dtype)) {
      out.push_back(in[i]);
      continue;
    }
    const Layer &s = in[i], &e1 = in[i + 1], &e3 = in[i + 2];
    Layer f = s;
    f.type = L_FIRE;
    f.name = s.name.substr(0, s.name.find('/'));
    f.out_buf = e1.out_buf;
    f.cout = e1.cout + e3.cout;
    f.fs = s.cout; f.fe1 = e1.cout; f.fe3 = e3.cout;
    f.kp_s = s.kparam; f.bp_s = s.bparam; f.kp_1 = e1.kparam; f.bp_1 = e1.bparam; f.kp_3 = e3.kparam; f.bp_3 = e3.bparam;
    f.y_cstride = f.cout; f.y_coffset = 0;
    f.flops = s.flops + e1.flops + e3.flops;
    // algorithmic bytes: fire input + concat output + the three weight sets (squeeze tensor not counted)
    const double npix = (double)net->batch * s.h * s.w;
    f.bytes = (npix * s.cin + npix * f.cout + (double)s.cin * s.cout + (double)s.cout * e1.cout + 9.0 * s.cout * e3.cout) *
                  (double)esz + 4.0 * (s.cout + e1.cout + e3.cout);
    out.push_back(f);
    i += 2;
  }
  net->layers.swap(out);
}

// fire module + the 3x3/s2 SAME max-pool behind it -> one launch of the streaming kernel's POOL form (fire3+pool3,
// fire5+pool5 of SqueezeDet): the module's full-resolution output never reaches HBM.  "fire_fuse" = 4 keeps them apart.
void fuse_fire_pools(sqdet_net* net, size_t esz) {
  if (conv_algo() != 0 || tune(3) == 2 || tune(3) == 3 || tune(3) == 4) return;
  std::vector<Layer> out;
  std::vector<Layer> in = net->layers;
  bool fused_any = false;
  for (size_t i = 0; i < in.size(); ++i) {
    const bool ok = i + 1 < in.size() && in[i].type == L_FIRE && in[i + 1].type == L_POOL && in[i + 1].k == 3 &&
                    in[i + 1].stride == 2 && in[i + 1].pad_mode == SQDET_PAD_SAME && in[i + 1].in_buf == in[i].out_buf &&
                    fire_stream_eligible(in[i].cin, in[i].fs, in[i].fe1, in[i].fe3, net->dtype);
    if (!ok) { out.push_back(in[i]); continue; }
    const Layer& p = in[i + 1];
    Layer f = in[i];
    f.fire_pool = 1;
    f.name = in[i].name + "+" + p.name;
    // The pooled tensor goes where the module's own output would have gone -- NOT into the pool's output buffer:
    // that is the ping-pong buffer the module READS (other workgroups are still reading it).  One ping-pong step
    // disappears, so the two buffers swap roles for every later layer.
    f.ho = p.ho; f.wo = p.wo;
    for (size_t k = i + 2; k < in.size(); ++k) {
      auto sw = [](int b) { return b == BUF_A ? BUF_B : (b == BUF_B ? BUF_A : b); };
      in[k].in_buf = sw(in[k].in_buf);
      in[k].out_buf = sw(in[k].out_buf);
    }
    fused_any = true;
    // algorithmic bytes: fire input + POOLED output + the three weight sets
    f.bytes = in[i].bytes - (double)net->batch * in[i].h * in[i].w * in[i].cout * (double)esz +
              (double)net->batch * p.ho * p.wo * in[i].cout * (double)esz;
    out.push_back(f);
    ++i;
  }
  net->layers.swap(out);
  if (fused_any) {   // either buffer may now hold what the other was sized for
    const size_t m = net->buf_elems[BUF_A] > net->buf_elems[BUF_B] ? net->buf_elems[BUF_A] : net->buf_elems[BUF_B];
    net->buf_elems[BUF_A] = net->buf_elems[BUF_B] = m;
  }
}

// Runs of consecutive fire modules (SqueezeDet: fire2 .. fire11, nets/squeezeDet.py:46-69): the only reader of a module's
// concat tensor -- pooled or not -- is the next module's squeeze1x1, so inside a run only 16-96-channel SQUEEZE tensors
// travel between the launches (float16).  Per member, by what covers its shape:
//   first module (input x):   one streaming launch, whole module + next squeeze (L_FIRESQ)  |  squeeze conv, then as below
//   module from its squeeze:  streaming launch expand (+ pool) + next squeeze (L_EXPSQ: one-chunk squeezes)
//                             |  ring chain launch expand + next squeeze (L_CHAIN; no pool)
//   last module:              expand + pool from the squeeze tensor (L_EXPAND)  |  chain launch writing the concat tensor
// "fire_fuse" = 5 keeps one launch per module, 6 chains the late (<= 100000 pixel) maps only, 7 never uses L_FIRESQ,
// 8 never uses L_EXPSQ (a pooled module then ends its run).
void fuse_chains(sqdet_net* net, size_t esz) {
  if (conv_algo() != 0 || tune(3) == 2 || tune(3) == 5) return;
  const std::vector<Layer> in = net->layers;
  std::vector<Layer> out;
  const int dt = net->dtype;
  auto base = [](const Layer& f) { return f.name.substr(0, f.name.find('+')); };
  // how member k of a run could run when it has a successor (0 = it cannot: the run ends before / at it)
  // the first module right behind a fused stem: its squeeze1x1 moves INTO the stem launch (fuse_stem_squeeze below) when the
  // module itself can then run from its squeeze tensor ("fire_fuse" = 9: not)
  auto stem_takes_squeeze = [&](const Layer& f, const Layer& nx) {
    return tune(3) != 9 && tune(3) != 8 && in.size() > 1 && &f == &in[1] && in[0].type == L_STEM && f.in_buf == in[0].out_buf && !f.fire_pool &&
           stem_squeeze_eligible(in[0].h, in[0].w, in[0].cout, in[0].k, in[0].pad_mode, in[0].pool_pad_mode, f.fs, dt, net->batch) &&
           fire_expand_squeeze_next_eligible(f.fs, f.fe1, f.fe3, nx.fs, 0, dt);
  };
  auto mid_impl = [&](const Layer& f, const Layer& nx, bool first) -> int {
    if (first && tune(3) != 7 && !f.fire_pool && !stem_takes_squeeze(f, nx) &&
        fire_squeeze_next_eligible(f.cin, f.fs, f.fe1, f.fe3, nx.fs, dt)) return L_FIRESQ;
    if (tune(3) != 8 && fire_expand_squeeze_next_eligible(f.fs, f.fe1, f.fe3, nx.fs, f.fire_pool, dt)) return L_EXPSQ;
    if (!f.fire_pool && fire_chain_eligible(f.fs, f.fe1, f.fe3, nx.fs, dt)) return L_CHAIN;
    return 0;
  };
  auto last_impl = [&](const Layer& f) -> int {
    if (f.fire_pool) return fire_expand_stream_eligible(f.fs, f.fe1, f.fe3, dt) ? L_EXPAND : 0;
    return fire_chain_eligible(f.fs, f.fe1, f.fe3, 0, dt) ? L_CHAIN : 0;
  };
  for (size_t i = 0; i < in.size();) {
    if (in[i].type != L_FIRE) { out.push_back(in[i]); ++i; continue; }
    // candidate run: consecutive fire modules, each reading its predecessor's (possibly pooled) output
    size_t j = i + 1;
    while (j < in.size() && in[j].type == L_FIRE && in[j].in_buf == in[j - 1].out_buf && in[j].h == in[j - 1].ho && in[j].w == in[j - 1].wo) ++j;
    if (tune(3) == 6) {   // late maps only
      if ((long)net->batch * in[i].h * in[i].w > 100000) { out.push_back(in[i]); ++i; continue; }
    }
    // longest prefix [i, e) every member of which has an implementation and whose final member is implemented as a LAST one
    size_t e = i;
    std::vector<int> impl;
    bool ended_with_last = false;
    for (size_t k = i; k < j; ++k) {
      const int mi = k + 1 < j ? mid_impl(in[k], in[k + 1], k == i) : 0;
      if (mi) { impl.push_back(mi); e = k + 1; continue; }
      const int li = last_impl(in[k]);     // k ends the run: the candidate's last member, or nothing carries it further
      if (li) { impl.push_back(li); e = k + 1; ended_with_last = true; }
      break;
    }
    while (!impl.empty() && !ended_with_last) {   // the successor of a "mid" member cannot run at all: that member ends the run
      const int li = last_impl(in[e - 1]);
      if (li) { impl.back() = li; ended_with_last = true; }
      else { impl.pop_back(); --e; }
    }
    if (e - i < 2) { out.push_back(in[i]); ++i; continue; }
    int sbuf = BUF_S;      // where the current member's squeeze tensor lives
    auto note_s = [&](const Layer& f, int ch, bool pooled) {
      const size_t el = (size_t)net->batch * (pooled ? f.ho : f.h) * (pooled ? f.wo : f.w) * (size_t)ch;
      if (el > net->buf_elems[BUF_S]) net->buf_elems[BUF_S] = el;
      if (el > net->buf_elems[BUF_T]) net->buf_elems[BUF_T] = el;
    };
    for (size_t k = i; k < e; ++k) {
      const Layer& f = in[k];
      const bool last = k + 1 == e;
      const int im = impl[k - i];
      const double npix = (double)net->batch * f.h * f.w;
      const double npix_out = (double)net->batch * f.ho * f.wo;
      const double wexp = (double)f.fs * f.fe1 + 9.0 * f.fs * f.fe3;
      if (k == i && im != L_FIRESQ) {     // the first module's squeeze as a plain conv
        Layer sq = f;
        sq.type = L_CONV;
        sq.fire_pool = 0;
        sq.name = base(f) + "/squeeze1x1";
        sq.out_buf = BUF_S;
        sq.cout = f.fs; sq.k = 1; sq.stride = 1; sq.pad_mode = SQDET_PAD_SAME; sq.relu = 1;
        sq.ho = f.h; sq.wo = f.w;
        sq.y_cstride = f.fs; sq.y_coffset = 0;
        sq.kparam = f.kp_s; sq.bparam = f.bp_s;
        sq.flops = 2.0 * f.cin * f.fs * npix;
        sq.bytes = (npix * f.cin + npix * f.fs + (double)f.cin * f.fs) * (double)esz + 4.0 * f.fs;
        note_s(f, f.fs, false);
        out.push_back(sq);
      }
      Layer c = f;
      c.type = im;
      const int fs2 = last ? 0 : in[k + 1].fs;
      c.fs2 = fs2;
      c.kp_s2 = last ? -1 : in[k + 1].kp_s;
      c.bp_s2 = last ? -1 : in[k + 1].bp_s;
      const std::string nxs = last ? std::string() : "+" + base(in[k + 1]) + "/squeeze1x1";
      const std::string pool_s = f.fire_pool ? f.name.substr(f.name.find('+')) : std::string();
      if (im == L_FIRESQ) {
        c.out_buf = BUF_S;
        c.name = base(f) + nxs;
        c.flops = f.flops + 2.0 * (f.fe1 + f.fe3) * fs2 * npix;
        c.bytes = (npix * f.cin + npix * fs2 + (double)f.cin * f.fs + wexp + (double)(f.fe1 + f.fe3) * fs2) * (double)esz +
                  4.0 * (f.fs + f.fe1 + f.fe3 + fs2);
        note_s(f, fs2, false);
        out.push_back(c);
        sbuf = BUF_S;
        continue;
      }
      c.in_buf = sbuf;
      if (im == L_EXPAND) {     // (last member with a pool: its pooled concat tensor goes where the fused layer's went)
        c.name = base(f) + "/expand" + pool_s;
        c.flops = (2.0 * f.fs * f.fe1 + 18.0 * f.fs * f.fe3) * npix;
        c.bytes = (npix * f.fs + npix_out * (f.fe1 + f.fe3) + wexp) * (double)esz + 4.0 * (f.fe1 + f.fe3);
        out.push_back(c);
        continue;
      }
      c.out_buf = last ? f.out_buf : (sbuf == BUF_S ? BUF_T : BUF_S);
      c.name = base(f) + "/expand" + pool_s + nxs;
      c.flops = (2.0 * f.fs * f.fe1 + 18.0 * f.fs * f.fe3) * npix + 2.0 * (f.fe1 + f.fe3) * fs2 * npix_out;
      // algorithmic bytes: squeeze tensor in + (next squeeze tensor | concat tensor) out + the weights
      c.bytes = (npix * f.fs + npix_out * (last ? f.fe1 + f.fe3 : fs2) + wexp + (double)(f.fe1 + f.fe3) * fs2) * (double)esz +
                4.0 * (f.fe1 + f.fe3 + fs2);
      if (im == L_CHAIN) {
        c.chain_off = net->param_bytes;
        net->param_bytes = align_up(net->param_bytes + sqdet_fire_chain_stream_bytes(f.fs, f.fe1, f.fe3, fs2, dt), 256);
      }
      if (!last) note_s(f, fs2, f.fire_pool != 0);
      out.push_back(c);
      sbuf = c.out_buf;
    }
    i = e;
  }
  net->layers.swap(out);
}

// expand1x1 + expand3x3 of a fire module that stayed three convs (SqueezeDet+'s squeeze depths 192 / 384: no fused fire kernel, no
// chain) -> one L_EXPAND launch of the tile kernel's PAIR form where it covers the shape (conv3x3_pair_eligible): both expands from ONE
// staged squeeze tile.  "fire_fuse" = 2 / 11: not.
void fuse_expand_pairs(sqdet_net* net, size_t esz) {
  if (conv_algo() != 0 || tune(3) == 2 || tune(3) == 11) return;
  std::vector<Layer> out;
  const std::vector<Layer>& in = net->layers;
  for (size_t i = 0; i < in.size(); ++i) {
    const bool pair = i + 1 < in.size() && in[i].type == L_CONV && in[i + 1].type == L_CONV && in[i].k == 1 && in[i + 1].k == 3 &&
                      in[i].in_buf == BUF_S && in[i + 1].in_buf == BUF_S && in[i].out_buf == in[i + 1].out_buf && in[i].stride == 1 &&
                      in[i + 1].stride == 1 && in[i].relu && in[i + 1].relu && in[i].cin == in[i + 1].cin && in[i].y_coffset == 0 &&
                      in[i + 1].y_coffset == in[i].cout && in[i].y_cstride == in[i].cout + in[i + 1].cout &&
                      in[i + 1].y_cstride == in[i].y_cstride && in[i].fold < 0 && in[i + 1].fold < 0 && !in[i].accum && !in[i + 1].accum &&
                      in[i + 1].pad_mode == SQDET_PAD_SAME &&
                      conv3x3_pair_eligible(net->batch, in[i].h, in[i].w, in[i].cin, in[i].cout, in[i + 1].cout, net->dtype);
    if (!pair) { out.push_back(in[i]); continue; }
    const Layer &e1 = in[i], &e3 = in[i + 1];
    Layer f = e3;
    f.type = L_EXPAND;
    f.name = e1.name.substr(0, e1.name.find('/')) + "/expand";
    f.fire_pool = 0;
    f.fs = e1.cin; f.fe1 = e1.cout; f.fe3 = e3.cout;
    f.kp_1 = e1.kparam; f.bp_1 = e1.bparam; f.kp_3 = e3.kparam; f.bp_3 = e3.bparam;
    f.cout = e1.cout + e3.cout;
    f.y_cstride = f.cout; f.y_coffset = 0;
    f.flops = e1.flops + e3.flops;
    // algorithmic bytes: squeeze tensor in (once) + concat tensor out + both weight sets
    const double npix = (double)net->batch * e1.h * e1.w;
    f.bytes = (npix * e1.cin + npix * f.cout + 10.0 * e1.cin * e1.cout) * (double)esz + 4.0 * f.cout;
    out.push_back(f);
    ++i;
  }
  net->layers.swap(out);
}

// L_STEM followed by the first chained module's squeeze1x1 (a plain 64 -> 16 conv emitted by fuse_chains) -> one L_STEMSQ
// launch: pool1's tensor is never written.
void fuse_stem_squeeze(sqdet_net* net, size_t esz) {
  if (conv_algo() != 0 || tune(3) == 9 || net->layers.size() < 3) return;
  const Layer& st = net->layers[0];
  const Layer& sq = net->layers[1];
  if (st.type != L_STEM || sq.type != L_CONV || sq.k != 1 || sq.stride != 1 || !sq.relu || sq.in_buf != st.out_buf) return;
  if (sq.cin != st.cout || sq.y_cstride != sq.cout || sq.y_coffset != 0 || sq.accum || sq.fold >= 0) return;
  if (net->layers[2].type != L_EXPSQ || net->layers[2].in_buf != sq.out_buf) return;
  if (!stem_squeeze_eligible(st.h, st.w, st.cout, st.k, st.pad_mode, st.pool_pad_mode, sq.cout, net->dtype, net->batch)) return;
  Layer f = st;
  f.type = L_STEMSQ;
  f.name = st.name + "+" + sq.name;
  f.out_buf = sq.out_buf;
  f.fs2 = sq.cout;
  f.kp_s2 = sq.kparam; f.bp_s2 = sq.bparam;
  f.flops = st.flops + sq.flops;
  // algorithmic bytes: input + squeeze tensor + both weight sets
  f.bytes = ((double)net->batch * st.h * st.w * 3 + (double)net->batch * st.ho * st.wo * sq.cout + (double)st.k * st.k * 3 * st.cout +
             (double)st.cout * sq.cout) * (double)esz + 4.0 * (st.cout + sq.cout);
  net->layers.erase(net->layers.begin() + 1);
  net->layers[0] = f;
}

// conv1 + pool1 -> one L_STEM launch when the fused kernel applies (decided at plan creation).
void fuse_stem(sqdet_net* net, size_t esz) {
  if (conv_algo() != 0 || net->layers.size() < 2) return;
  Layer& c = net->layers[0];
  const Layer& p = net->layers[1];
  if (c.type != L_CONV || p.type != L_POOL || c.cin != 3 || c.stride != 2 || !c.relu) return;
  if (!((c.k == 3 && c.cout == 64) || (c.k == 7 && (c.cout == 96 || c.cout == 64)))) return;
  if (p.k != 3 || p.stride != 2 || p.in_buf != c.out_buf) return;
  Layer f = c;
  f.type = L_STEM;
  f.name = c.name + "+" + p.name;
  f.out_buf = p.out_buf;
  f.pool_pad_mode = p.pad_mode;
  f.ho = p.ho; f.wo = p.wo;
  f.y_cstride = c.cout; f.y_coffset = 0;
  // algorithmic bytes: input + POOLED output + weights (the conv activations never reach HBM)
  f.bytes = ((double)net->batch * c.h * c.w * 3 + (double)net->batch * p.ho * p.wo * c.cout +
             (double)c.k * c.k * 3 * c.cout) * (double)esz + c.cout * 4.0;
  net->layers.erase(net->layers.begin() + 1);
  net->layers[0] = f;
}

}  // namespace

extern "C" int sqdet_net_create(sqdet_net_t** out, int arch, int dtype, int batch, int img_h, int img_w, int classes,
                                int anchors_per_grid) {
  SQDET_REQUIRE(out, "net_create: null out");
  SQDET_REQUIRE(arch == SQDET_ARCH_SQUEEZEDET || arch == SQDET_ARCH_SQUEEZEDET_PLUS || arch == SQDET_ARCH_RESNET50,
                "net_create: bad arch %d", arch);
  SQDET_REQUIRE(dtype == SQDET_F16 || dtype == SQDET_F32, "net_create: bad dtype %d", dtype);
  SQDET_REQUIRE(batch > 0 && img_h >= 64 && img_w >= 64 && classes > 0 && anchors_per_grid > 0, "net_create: bad dims");
  sqdet_net* net = new sqdet_net();
  net->arch = arch; net->dtype = dtype; net->batch = batch; net->img_h = img_h; net->img_w = img_w;
  net->classes = classes; net->apg = anchors_per_grid;
  Builder b;
  b.net = net; b.h = img_h; b.w = img_w; b.c = 3; b.cur = BUF_INPUT; b.esz = dtype_size(dtype);
  const int nout = anchors_per_grid * (classes + 1 + 4);  // nets/squeezeDet.py:76
  if (arch == SQDET_ARCH_SQUEEZEDET) {
    const FireSpec* f = kSqueezeDetFires;
    b.conv_layer("conv1", 64, 3, 2, SQDET_PAD_SAME, 1, false);
    b.pool_layer("pool1", 3, 2, SQDET_PAD_SAME);
    b.fire_layer(f[0]); b.fire_layer(f[1]);
    b.pool_layer("pool3", 3, 2, SQDET_PAD_SAME);
    b.fire_layer(f[2]); b.fire_layer(f[3]);
    b.pool_layer("pool5", 3, 2, SQDET_PAD_SAME);
    for (int i = 4; i < 10; ++i) b.fire_layer(f[i]);
  } else if (arch == SQDET_ARCH_RESNET50) {
    // resnet50_convDet.py:41-118: conv1 7x7/2 (+bias, BN) -> pool1 3x3/2 VALID -> res2a..res4f
    b.conv_bn("conv1", BUF_INPUT, BUF_A, 3, 64, 7, 2, 1, true, 0);
    b.h = net->layers.back().ho; b.w = net->layers.back().wo; b.c = 64; b.cur = BUF_A;
    b.pool_layer("pool1", 3, 2, SQDET_PAD_VALID);
    b.res_block("conv2_x", "2a", 64, 256, false, true);
    b.res_block("conv2_x", "2b", 64, 256, false, false);
    b.res_block("conv2_x", "2c", 64, 256, false, false);
    b.res_block("conv3_x", "3a", 128, 512, true, true);
    for (const char* n : {"3b", "3c", "3d"}) b.res_block("conv3_x", n, 128, 512, false, false);
    b.res_block("conv4_x", "4a", 256, 1024, true, true);
    for (const char* n : {"4b", "4c", "4d", "4e", "4f"}) b.res_block("conv4_x", n, 256, 1024, false, false);
  } else {
    const FireSpec* f = kSqueezeDetPlusFires;
    b.conv_layer("conv1", 96, 7, 2, SQDET_PAD_VALID, 1, false);
    b.pool_layer("pool1", 3, 2, SQDET_PAD_VALID);
    b.fire_layer(f[0]); b.fire_layer(f[1]); b.fire_layer(f[2]);
    b.pool_layer("pool4", 3, 2, SQDET_PAD_VALID);
    b.fire_layer(f[3]); b.fire_layer(f[4]); b.fire_layer(f[5]); b.fire_layer(f[6]);
    b.pool_layer("pool8", 3, 2, SQDET_PAD_VALID);
    b.fire_layer(f[7]); b.fire_layer(f[8]); b.fire_layer(f[9]);
  }
  // dropout11 / drop4 is the identity at inference (keep_prob = 1.0, nn_skeleton.py:78)
  b.conv_layer(arch == SQDET_ARCH_RESNET50 ? "conv5" : "conv12", nout, 3, 1, SQDET_PAD_SAME, 0, true);
  net->gh = b.h; net->gw = b.w; net->out_ch = nout;
  fuse_stem(net, b.esz);
  fuse_fires(net, b.esz);
  fuse_fire_pools(net, b.esz);
  fuse_chains(net, b.esz);
  fuse_expand_pairs(net, b.esz);
  fuse_stem_squeeze(net, b.esz);
  net->fold_scratch_off = net->param_bytes;
  net->param_bytes = align_up(net->param_bytes + net->fold_scratch_bytes, 256);
  size_t off = 0;
  for (int i = 0; i < NUM_BUFS; ++i) {
    net->buf_off[i] = off;
    off = align_up(off + net->buf_elems[i] * b.esz, 256);
  }
  net->workspace_bytes = off;
  *out = net;
  return SQDET_OK;
}

extern "C" void sqdet_net_destroy(sqdet_net_t* net) {
  if (!net) return;
  for (hipEvent_t e : net->events) (void)hipEventDestroy(e);
  for (hipEvent_t e : net->probe_events) (void)hipEventDestroy(e);
  delete net;
}

extern "C" int sqdet_net_num_params(const sqdet_net_t* net) { return net ? (int)net->params.size() : 0; }

extern "C" int sqdet_net_param_info(const sqdet_net_t* net, int index, char* name, size_t name_cap, int shape[4],
                                    int* ndim) {
  SQDET_REQUIRE(net && index >= 0 && index < (int)net->params.size(), "param_info: bad index");
  const Param& p = net->params[index];
  if (name && name_cap) {
    strncpy(name, p.name.c_str(), name_cap - 1);
    name[name_cap - 1] = 0;
  }
  if (shape) for (int i = 0; i < 4; ++i) shape[i] = p.shape[i];
  if (ndim) *ndim = p.ndim;
  return SQDET_OK;
}

extern "C" size_t sqdet_net_param_bytes(const sqdet_net_t* net) { return net ? net->param_bytes : 0; }
extern "C" size_t sqdet_net_workspace_bytes(const sqdet_net_t* net) { return net ? net->workspace_bytes : 0; }

extern "C" int sqdet_net_bind(sqdet_net_t* net, void* param_mem, void* workspace_mem) {
  SQDET_REQUIRE(net && param_mem && workspace_mem, "net_bind: null pointer");
  SQDET_REQUIRE(((uintptr_t)param_mem % 256) == 0 && ((uintptr_t)workspace_mem % 256) == 0,
                "net_bind: buffers must be 256-byte aligned");
  net->param_mem = reinterpret_cast<char*>(param_mem);
  net->workspace = reinterpret_cast<char*>(workspace_mem);
  return SQDET_OK;
}

extern "C" int sqdet_net_set_param(sqdet_net_t* net, const char* name, const float* value_f32, sqdet_stream_t stream) {
  SQDET_REQUIRE(net && name && value_f32, "net_set_param: null pointer");
  if (!net->param_mem) { set_error("net_set_param: call sqdet_net_bind first"); return SQDET_ESTATE; }
  for (const Param& p : net->params) {
    if (p.name != name) continue;
    if (p.fold >= 0) {  // _conv_bn_layer: keep the float32 value, fold + pack lazily
      BnFold& f = net->folds[p.fold];
      const bool kernel = p.ndim == 4;
      SQDET_CHECK_HIP(hipMemcpyAsync(net->param_mem + (kernel ? f.raw_off : p.offset), value_f32,
                                     kernel ? (size_t)f.k * f.k * f.cin * f.cout * 4 : (size_t)p.shape[0] * 4,
                                     hipMemcpyDeviceToDevice, as_stream(stream)));
      f.dirty = true;
      return SQDET_OK;
    }
    if (p.ndim == 4) {
      // kernels that also travel in a chain stream (expand1x1 / expand3x3 of a chained module, squeeze1x1 of its successor)
      const int pi = (int)(&p - net->params.data());
      for (const Layer& L : net->layers) {
        if (L.type != L_CHAIN || (pi != L.kp_1 && pi != L.kp_3 && pi != L.kp_s2)) continue;
        const int rc = sqdet_fire_chain_pack(pi == L.kp_1 ? value_f32 : nullptr, pi == L.kp_3 ? value_f32 : nullptr,
                                             pi == L.kp_s2 ? value_f32 : nullptr, net->param_mem + L.chain_off, L.fs, L.fe1,
                                             L.fe3, L.fs2, net->dtype, stream);
        if (rc != SQDET_OK) return rc;
      }
      return sqdet_conv_pack_weights(value_f32, net->param_mem + p.offset, p.shape[0], p.shape[2], p.shape[3],
                                     net->dtype, stream);
    }
    SQDET_CHECK_HIP(hipMemcpyAsync(net->param_mem + p.offset, value_f32, (size_t)p.shape[0] * 4,
                                   hipMemcpyDeviceToDevice, as_stream(stream)));
    return SQDET_OK;
  }
  set_error("net_set_param: no parameter named '%s'", name);
  return SQDET_EINVAL;
}

extern "C" int sqdet_net_set_bn_epsilon(sqdet_net_t* net, float eps) {
  SQDET_REQUIRE(net && eps >= 0.f, "net_set_bn_epsilon: bad arguments");
  net->bn_eps = eps;
  for (BnFold& f : net->folds) f.dirty = true;
  return SQDET_OK;
}

extern "C" int sqdet_net_output_dims(const sqdet_net_t* net, int* gh, int* gw, int* channels) {
  SQDET_REQUIRE(net, "net_output_dims: null net");
  if (gh) *gh = net->gh;
  if (gw) *gw = net->gw;
  if (channels) *channels = net->out_ch;
  return SQDET_OK;
}

extern "C" int sqdet_net_forward(sqdet_net_t* net, const void* image_input, void* preds, sqdet_stream_t stream) {
  SQDET_REQUIRE(net && image_input && preds, "net_forward: null pointer");
  if (!net->param_mem || !net->workspace) { set_error("net_forward: call sqdet_net_bind first"); return SQDET_ESTATE; }
  hipStream_t st = as_stream(stream);
  const int frc = refresh_folds(net, st);
  if (frc != SQDET_OK) return frc;
  const int nl = (int)net->layers.size();
  auto one = [&](int i, int n0, int nb, hipStream_t ls) -> int {
    const bool probe = n0 == 0 && i == net->probe_layer && 2 * (net->probe_count + 1) <= (int)net->probe_events.size();
    if (probe) SQDET_CHECK_HIP(hipEventRecord(net->probe_events[2 * net->probe_count], ls));
    const int rc = run_layer_part(net, net->layers[i], image_input, preds, n0, nb, ls);
    if (rc != SQDET_OK) return rc;
    if (probe) {
      SQDET_CHECK_HIP(hipEventRecord(net->probe_events[2 * net->probe_count + 1], ls));
      ++net->probe_count;
    }
    return SQDET_OK;
  };
  for (int i = 0; i < nl; ++i) {
    if (i == net->signal_layer && net->signal_event) SQDET_CHECK_HIP(hipEventRecord(net->signal_event, st));
    const int rc = one(i, 0, net->batch, st);
    if (rc != SQDET_OK) { net->job_set = false; return rc; }
  }
  net->job_set = false;      // (one-shot: its riders went out with this forward's fire_chain launches)
  return SQDET_OK;
}

static int net_rider_capacity(const sqdet_net* net) {
  int cap = 0;
  for (const Layer& L : net->layers)
    if (L.type == L_CHAIN) cap += layer_riders(net, L);
  return cap;
}

extern "C" int sqdet_net_rider_capacity(const sqdet_net_t* net) { return net ? net_rider_capacity(net) : 0; }

extern "C" int sqdet_net_set_post_job(sqdet_net_t* net, const void* preds, const float* scores, const float* anchors, float* out_boxes,
                                      float* out_probs, int32_t* out_cls, int32_t* out_index, int32_t* out_count, int n, int gh,
                                      int gw, int apg, int classes, float img_w, float img_h, float exp_thresh, int top_n,
                                      int max_out, double nms_thresh, int dtype) {
  SQDET_REQUIRE(net, "net_set_post_job: null net");
  if (!preds) { net->job_set = false; return SQDET_OK; }      // cancel
  SQDET_REQUIRE(scores && anchors && out_boxes && out_probs && out_cls && out_index && out_count, "net_set_post_job: null pointer");
  SQDET_REQUIRE(n > 0 && gh > 0 && gw > 0 && apg > 0 && classes > 0 && max_out >= top_n, "net_set_post_job: bad dims");
  const int A = gh * gw * apg;
  SQDET_UNSUPPORTED(dtype != SQDET_F16 || !(top_n > 0 && top_n <= 64 && top_n < A && A <= 20480),
                    "net_set_post_job: float16 preds and the top-N branch (0 < top_n <= 64 < anchors <= 20480) only");
  SQDET_UNSUPPORTED(n > net_rider_capacity(net), "net_set_post_job: %d images, but this plan's fire_chain launches leave %d CUs idle",
                    n, net_rider_capacity(net));
  sqdet::FilterArgs& a = net->job_fa;
  a.boxes = nullptr; a.probs = scores; a.cls = nullptr;
  a.out_boxes = out_boxes; a.out_probs = out_probs; a.out_cls = out_cls; a.out_index = out_index; a.out_count = out_count;
  a.A = A; a.C = classes; a.top_n = top_n; a.max_out = max_out; a.cap = 0; a.use_topn = 1;
  a.nms_thresh = nms_thresh; a.prob_thresh = 0.f;
  sqdet::DecodeArgs& d = net->job_da;
  d.preds = preds; d.anchors = anchors; d.cells = gh * gw; d.apg = apg; d.C = classes; d.dtype = dtype;
  d.w1 = img_w - 1.0f; d.h1 = img_h - 1.0f; d.thr = exp_thresh; d.slope = (float)exp((double)exp_thresh);
  net->job_n = n;
  net->job_set = true;
  return SQDET_OK;
}

extern "C" int sqdet_net_set_signal(sqdet_net_t* net, int layer_index, void* hip_event) {
  SQDET_REQUIRE(net, "net_set_signal: null net");
  SQDET_REQUIRE(!hip_event || (layer_index >= 0 && layer_index < (int)net->layers.size()), "net_set_signal: bad layer index %d", layer_index);
  net->signal_layer = hip_event ? layer_index : -1;
  net->signal_event = reinterpret_cast<hipEvent_t>(hip_event);
  return SQDET_OK;
}

extern "C" int sqdet_net_overlap_layer(const sqdet_net_t* net) {
  if (!net) return -1;
  for (size_t i = 0; i < net->layers.size(); ++i)
    if (net->layers[i].type == L_CHAIN) return (int)i;
  return -1;
}

static bool net_scores_ok(const sqdet_net* net) {
  if (net->layers.empty()) return false;
  const Layer& L = net->layers.back();
  return L.type == L_CONV && L.k == 3 && L.stride == 1 && L.pad_mode == SQDET_PAD_SAME && !L.relu && !L.accum && L.fold < 0 &&
         L.out_buf == BUF_PREDS && L.cout == net->apg * (net->classes + 5) && L.y_cstride == L.cout && L.y_coffset == 0 &&
         sqdet_convdet_scores_supported(L.cin, net->apg, net->classes, net->dtype) != 0;
}

extern "C" int sqdet_net_scores_supported(const sqdet_net_t* net) { return net && net_scores_ok(net) ? 1 : 0; }

extern "C" int sqdet_net_set_scores(sqdet_net_t* net, float* scores) {
  SQDET_REQUIRE(net, "net_set_scores: null net");
  SQDET_UNSUPPORTED(scores && !net_scores_ok(net), "net_set_scores: this plan's last layer has no score epilogue (float16 SqueezeDet-style ConvDet head only)");
  net->scores = scores;
  return SQDET_OK;
}

extern "C" int sqdet_net_set_probe(sqdet_net_t* net, int layer_index, int max_records) {
  SQDET_REQUIRE(net && layer_index >= -1 && layer_index < (int)net->layers.size() && max_records >= 0,
                "net_set_probe: bad arguments");
  net->probe_layer = layer_index;
  net->probe_count = 0;
  while ((int)net->probe_events.size() < 2 * max_records) {
    hipEvent_t e;
    SQDET_CHECK_HIP(hipEventCreate(&e));
    net->probe_events.push_back(e);
  }
  return SQDET_OK;
}

extern "C" int sqdet_net_read_probe(sqdet_net_t* net, float* host_ms, int capacity, int* count) {
  SQDET_REQUIRE(net && host_ms && count, "net_read_probe: null pointer");
  int n = net->probe_count < capacity ? net->probe_count : capacity;
  for (int i = 0; i < n; ++i) {
    SQDET_CHECK_HIP(hipEventSynchronize(net->probe_events[2 * i + 1]));
    SQDET_CHECK_HIP(hipEventElapsedTime(&host_ms[i], net->probe_events[2 * i], net->probe_events[2 * i + 1]));
  }
  *count = n;
  net->probe_count = 0;
  return SQDET_OK;
}

extern "C" int sqdet_net_num_layers(const sqdet_net_t* net) { return net ? (int)net->layers.size() : 0; }

extern "C" int sqdet_net_layer_info(const sqdet_net_t* net, int index, char* name, size_t name_cap, double* flops,
                                    double* bytes) {
  SQDET_REQUIRE(net && index >= 0 && index < (int)net->layers.size(), "layer_info: bad index");
  const Layer& L = net->layers[index];
  if (name && name_cap) {
    strncpy(name, L.name.c_str(), name_cap - 1);
    name[name_cap - 1] = 0;
  }
  if (flops) *flops = L.flops;
  if (bytes) *bytes = L.bytes;
  return SQDET_OK;
}

extern "C" int sqdet_net_forward_timed(sqdet_net_t* net, const void* image_input, void* preds, float* host_ms,
                                       sqdet_stream_t stream) {
  SQDET_REQUIRE(net && image_input && preds && host_ms, "net_forward_timed: null pointer");
  if (!net->param_mem || !net->workspace) { set_error("net_forward_timed: call sqdet_net_bind first"); return SQDET_ESTATE; }
  const size_t nl = net->layers.size();
  while (net->events.size() < nl + 1) {
    hipEvent_t e;
    SQDET_CHECK_HIP(hipEventCreate(&e));
    net->events.push_back(e);
  }
  hipStream_t st = as_stream(stream);
  const int frc = refresh_folds(net, st);
  if (frc != SQDET_OK) return frc;
  SQDET_CHECK_HIP(hipEventRecord(net->events[0], st));
  for (size_t i = 0; i < nl; ++i) {
    const int rc = run_layer(net, net->layers[i], image_input, preds, st);
    if (rc != SQDET_OK) return rc;
    SQDET_CHECK_HIP(hipEventRecord(net->events[i + 1], st));
  }
  SQDET_CHECK_HIP(hipEventSynchronize(net->events[nl]));
  for (size_t i = 0; i < nl; ++i) SQDET_CHECK_HIP(hipEventElapsedTime(&host_ms[i], net->events[i], net->events[i + 1]));
  return SQDET_OK;
}

static int fire_fwd_impl(const void* x, const void* w_s, const float* b_s, const void* w_e1, const float* b_e1,
                         const void* w_e3, const float* b_e3, void* sq_scratch, void* y, int n, int h, int w,
                         int cin, int s1x1, int e1x1, int e3x3, int dtype, bool keep, sqdet_stream_t stream);

extern "C" int sqdet_fire_fwd(const void* x, const void* w_s, const float* b_s, const void* w_e1, const float* b_e1,
                              const void* w_e3, const float* b_e3, void* sq_scratch, void* y, int n, int h, int w,
                              int cin, int s1x1, int e1x1, int e3x3, int dtype, sqdet_stream_t stream) {
  return fire_fwd_impl(x, w_s, b_s, w_e1, b_e1, w_e3, b_e3, sq_scratch, y, n, h, w, cin, s1x1, e1x1, e3x3, dtype, false, stream);
}

extern "C" int sqdet_fire_fwd_keep(const void* x, const void* w_s, const float* b_s, const void* w_e1, const float* b_e1,
                                   const void* w_e3, const float* b_e3, void* sq_out, void* y, int n, int h, int w,
                                   int cin, int s1x1, int e1x1, int e3x3, int dtype, sqdet_stream_t stream) {
  return fire_fwd_impl(x, w_s, b_s, w_e1, b_e1, w_e3, b_e3, sq_out, y, n, h, w, cin, s1x1, e1x1, e3x3, dtype, true, stream);
}

static int fire_fwd_impl(const void* x, const void* w_s, const float* b_s, const void* w_e1, const float* b_e1,
                         const void* w_e3, const float* b_e3, void* sq_scratch, void* y, int n, int h, int w,
                         int cin, int s1x1, int e1x1, int e3x3, int dtype, bool keep, sqdet_stream_t stream) {
  SQDET_REQUIRE(sq_scratch, "fire_fwd: null squeeze buffer");
  hipStream_t st = as_stream(stream);
  bool handled = false;
  int rc = SQDET_OK;
  if (tune(3) != 2) {  // one fused launch when eligible (keep: its squeeze epilogue also writes the squeeze tensor)
    rc = fire_fused_launch_keep(x, w_s, b_s, w_e1, b_e1, w_e3, b_e3, keep ? sq_scratch : nullptr, y, n, h, w, cin, s1x1, e1x1, e3x3,
                                dtype, st, &handled);
    if (rc != SQDET_OK || handled) return rc;
  }
  rc = conv2d_launch(x, w_s, b_s, sq_scratch, n, h, w, cin, s1x1, 1, 1, SQDET_PAD_SAME, 1, dtype, s1x1, 0, st);
  if (rc != SQDET_OK) return rc;
  rc = conv2d_launch(sq_scratch, w_e1, b_e1, y, n, h, w, s1x1, e1x1, 1, 1, SQDET_PAD_SAME, 1, dtype, e1x1 + e3x3, 0, st);
  if (rc != SQDET_OK) return rc;
  return conv2d_launch(sq_scratch, w_e3, b_e3, y, n, h, w, s1x1, e3x3, 3, 1, SQDET_PAD_SAME, 1, dtype, e1x1 + e3x3,
                       e1x1, st);
}

extern "C" int sqdet_fire_maxpool_fwd(const void* x, const void* w_s, const float* b_s, const void* w_e1, const float* b_e1,
                                      const void* w_e3, const float* b_e3, void* sq_scratch, void* fire_scratch, void* y,
                                      int n, int h, int w, int cin, int s1x1, int e1x1, int e3x3, int dtype,
                                      sqdet_stream_t stream) {
  SQDET_REQUIRE(fire_scratch && y, "fire_maxpool_fwd: null pointer");
  hipStream_t st = as_stream(stream);
  const int ff = tune(3);
  if (ff != 2 && ff != 3 && ff != 4) {   // one launch when the streaming kernel covers the shape (the scratches stay untouched)
    bool handled = false;
    const int rc = fire_stream_launch_ex(x, w_s, b_s, w_e1, b_e1, w_e3, b_e3, y, n, h, w, cin, s1x1, e1x1, e3x3, dtype, 1, st,
                                         &handled);
    if (rc != SQDET_OK || handled) return rc;
  }
  const int rc = sqdet_fire_fwd(x, w_s, b_s, w_e1, b_e1, w_e3, b_e3, sq_scratch, fire_scratch, n, h, w, cin, s1x1, e1x1, e3x3,
                                dtype, stream);
  if (rc != SQDET_OK) return rc;
  return maxpool_launch(fire_scratch, y, n, h, w, e1x1 + e3x3, 3, 2, SQDET_PAD_SAME, dtype, st);
}

extern "C" int sqdet_fire_expand_fwd(const void* sq_in, const void* w_e1, const float* b_e1, const void* w_e3, const float* b_e3,
                                     void* y, int n, int h, int w, int s1x1, int e1x1, int e3x3, int pool, int dtype,
                                     sqdet_stream_t stream) {
  SQDET_REQUIRE(sq_in && w_e1 && b_e1 && w_e3 && b_e3 && y, "fire_expand_fwd: null pointer");
  hipStream_t st = as_stream(stream);
  bool handled = false;
  int rc = fire_expand_stream_launch(sq_in, w_e1, b_e1, w_e3, b_e3, y, n, h, w, s1x1, e1x1, e3x3, dtype, pool, st, &handled);
  if (rc != SQDET_OK || handled) return rc;
  SQDET_UNSUPPORTED(pool != 0, "fire_expand_fwd: the pooled form needs a shape the streaming kernel covers");
  rc = conv3x3_pair_launch(sq_in, w_e3, b_e3, w_e1, b_e1, y, n, h, w, s1x1, e1x1, e3x3, dtype, st, &handled);   // deep squeezes: one tile launch
  if (rc != SQDET_OK || handled) return rc;
  rc = conv2d_launch(sq_in, w_e1, b_e1, y, n, h, w, s1x1, e1x1, 1, 1, SQDET_PAD_SAME, 1, dtype, e1x1 + e3x3, 0, st);
  if (rc != SQDET_OK) return rc;
  return conv2d_launch(sq_in, w_e3, b_e3, y, n, h, w, s1x1, e3x3, 3, 1, SQDET_PAD_SAME, 1, dtype, e1x1 + e3x3, e1x1, st);
}

extern "C" int sqdet_fire_expand_pair_supported(int n, int h, int w, int s1x1, int e1x1, int e3x3, int dtype) {
  return conv3x3_pair_eligible(n, h, w, s1x1, e1x1, e3x3, dtype) ? 1 : 0;
}

extern "C" int sqdet_fire_squeeze_next_fwd(const void* x, const void* w_s, const float* b_s, const void* w_e1, const float* b_e1,
                                           const void* w_e3, const float* b_e3, const void* w_next_s, const float* b_next_s,
                                           void* sq_out, int n, int h, int w, int cin, int s1x1, int e1x1, int e3x3,
                                           int next_s1x1, int dtype, sqdet_stream_t stream) {
  SQDET_REQUIRE(x && w_s && b_s && w_e1 && b_e1 && w_e3 && b_e3 && w_next_s && b_next_s && sq_out, "fire_squeeze_next_fwd: null pointer");
  bool handled = false;
  const int rc = fire_squeeze_next_launch(x, w_s, b_s, w_e1, b_e1, w_e3, b_e3, w_next_s, b_next_s, sq_out, n, h, w, cin, s1x1, e1x1,
                                          e3x3, next_s1x1, dtype, as_stream(stream), &handled);
  if (rc != SQDET_OK) return rc;
  SQDET_UNSUPPORTED(!handled, "fire_squeeze_next_fwd: shape not covered (see sqdet_fire_squeeze_next_supported)");
  return SQDET_OK;
}

extern "C" int sqdet_fire_squeeze_next_supported(int cin, int s1x1, int e1x1, int e3x3, int next_s1x1, int dtype) {
  return fire_squeeze_next_eligible(cin, s1x1, e1x1, e3x3, next_s1x1, dtype) ? 1 : 0;
}

extern "C" int sqdet_fire_expand_squeeze_next_supported(int s1x1, int e1x1, int e3x3, int next_s1x1, int pool, int dtype) {
  return fire_expand_squeeze_next_eligible(s1x1, e1x1, e3x3, next_s1x1, pool, dtype) ? 1 : 0;
}

extern "C" int sqdet_fire_expand_squeeze_next_fwd(const void* sq_in, const void* w_e1, const float* b_e1, const void* w_e3,
                                                  const float* b_e3, const void* w_next_s, const float* b_next_s, void* sq_out,
                                                  int n, int h, int w, int s1x1, int e1x1, int e3x3, int next_s1x1, int pool,
                                                  int dtype, sqdet_stream_t stream) {
  SQDET_REQUIRE(sq_in && w_e1 && b_e1 && w_e3 && b_e3 && w_next_s && b_next_s && sq_out, "fire_expand_squeeze_next_fwd: null pointer");
  bool handled = false;
  const int rc = fire_expand_squeeze_next_launch(sq_in, w_e1, b_e1, w_e3, b_e3, w_next_s, b_next_s, sq_out, n, h, w, s1x1, e1x1, e3x3,
                                                 next_s1x1, pool, dtype, as_stream(stream), &handled);
  if (rc != SQDET_OK) return rc;
  SQDET_UNSUPPORTED(!handled, "fire_expand_squeeze_next_fwd: shape not covered (see sqdet_fire_expand_squeeze_next_supported)");
  return SQDET_OK;
}

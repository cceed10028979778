// U-Net graph of rfi_toolbox/models/unet.py on top of the HIP kernels (host orchestration).
#pragma once
#include "planes.hpp"

namespace rfi {

struct DevBuf {
    rfi_ctx* ctx = nullptr;
    float* p = nullptr;
    size_t n = 0;
    void ensure(rfi_ctx* c, size_t floats);
    void free();
};

// a plane tensor (planes.hpp) owned by a model: [pixels][chunks][P][16] bf16 + a zeroed 64-byte tail
struct PlaneBuf {
    rfi_ctx* ctx = nullptr;
    bf16_t* p = nullptr;
    size_t elems = 0;
    int64_t pstride = 0;
    int nchunks = 0;
    void ensure(rfi_ctx* c, int64_t pixels, int C, int P);
    void free();
};

// Conv3x3(pad 1)+bias -> BatchNorm2d -> ReLU
struct ConvBN {
    std::string conv_name, bn_name;     // e.g. "encoder1.conv.conv.0", "encoder1.conv.conv.1"
    int cin = 0, cout = 0;
    int cin_p = 0;                      // cin rounded up to a multiple of 4 (library layout; zero padded)
    size_t w_off = 0, b_off = 0, g_off = 0, be_off = 0;   // offsets into the flat param/grad buffers
    size_t dbias_rec_off = 0;           // this layer's region of rfi_model::dbias_pool (floats)
    int ema_repeats = 1;
    int level = 1;                      // resolution level of the OUTPUT: H >> (level - 1)
    int R = 3, stride = 1;              // ResNet-style encoder: 3x3 stride 1 / 2, or the 1x1 stride-2 projection (R = 1)
    bool has_bias = true;               // false: Conv2d(bias=False) in front of a BatchNorm (the bias slot stays 0, no state_dict entry)
    float* ws2d = nullptr;              // stride 2: filters of the 2x2 form on the space-to-depth input [4][cout][4 cin]
    float* wds2d = nullptr;             // ... and their dgrad layout [4'][4 cin][cout]
    float* ws2d3 = nullptr;             // ... and the 3 x bf16 records of both (a launch without them splits a temporary copy
    float* wds2d3 = nullptr;            // and drains the stream to free it: six stalls per step)
    bool has_bn = true;                 // false: Conv+bias -> ReLU (SimpleCNN); scale=1, shift=0 stay fixed
    int64_t nbt = 0;                    // num_batches_tracked (host side)
    // device per-channel state: [running_mean | running_var | mean | invstd | scale | shift | c1 | c2]
    float* chan = nullptr;
    float* running_mean() const { return chan; }
    float* running_var() const { return chan + cout; }
    float* mean() const { return chan + 2 * cout; }
    float* invstd() const { return chan + 3 * cout; }
    float* scale() const { return chan + 4 * cout; }
    float* shift() const { return chan + 5 * cout; }
    float* c1() const { return chan + 6 * cout; }
    float* c2() const { return chan + 7 * cout; }
    float* wd = nullptr;                // dgrad-layout copy of the weight [9][cin_p][cout]
    float* w3 = nullptr;                // 3 x bf16 records of the forward layout (launch_weights_to_x3)
    float* wd3 = nullptr;               // ... and of the dgrad layout
    bf16_t* wBf = nullptr;              // plane kernels: filters in MFMA B-operand order, forward ...
    bf16_t* wBd = nullptr;              // ... and input-gradient direction (planes.hpp)
    bf16_t* ws3f = nullptr;             // wave-specialised float32 kernel (conv_ws.hip): B-operand order with three planes, one K
    bf16_t* ws3d = nullptr;             // segment; forward and input-gradient direction (3x3 stride-1 layers, channels % 16 == 0)
};

// ConvTranspose2d(k2,s2)+bias
struct UpConv {
    std::string name;                   // "decoder4.up"
    int cin = 0, cout = 0;
    size_t w_off = 0, b_off = 0;        // forward layout [4][cout][cin]
    size_t dbias_rec_off = 0;           // this layer's region of rfi_model::dbias_pool (floats)
    float* wd = nullptr;                // dgrad layout [4][cin][cout]
    float* w3 = nullptr;                // 3 x bf16 records of both layouts
    float* wd3 = nullptr;
    bf16_t* wBf = nullptr;              // plane kernels (bfloat16 flow): the four taps as ONE 1x1 contraction with 4 cout channels ...
    bf16_t* wBd = nullptr;              // ... and the 2x2 stride-2 contraction of the input gradient
};

struct Entry {
    std::string name;
    int ndim = 0;
    int64_t dims[4] = {0, 0, 0, 0};
    int kind = 0;        // 0 conv weight (OIHW), 1 convT weight (IOHW), 2 vector param, 3 running_mean,
                         // 4 running_var, 5 num_batches_tracked, 6 final weight, 8 / 9 frozen BatchNorm weight / bias (buffers), 7 1x1 conv weight of convs[layer]
                         // ([cout][cin][1][1] is the library's [1 tap][cout][cin] as is)
    int layer = -1;      // index into convs / ups; -1 for the head
    int which = 0;       // vector param: 0 conv bias, 1 bn gamma, 2 bn beta, 3 up bias, 4 head bias
    int64_t numel() const {
        int64_t n = 1;
        for (int i = 0; i < ndim; ++i) n *= dims[i];
        return n;
    }
};

}  // namespace rfi

struct rfi_model {
    rfi_ctx* ctx = nullptr;
    int in_ch = 0, out_ch = 0, feat = 0, depth = 0;
    int arch = 0;                     // 0: U-Net (models/unet.py), 1: 3-layer CNN (depth == 0),
                                      // 2: U-Net with a ResNet-18-style encoder (model_resnet.cpp),
                                      // 3: Mask R-CNN's per-RoI mask head (model_mask.cpp; depth = conv layers),
                                      // 4: RPN head (the same stack without the transposed conv),
                                      // 5: ResNet-50-FPN backbone, frozen BatchNorm (model_backbone.cpp), 6: FC box head (model_mlp.cpp)
    int i_bott = 0;                   // index of the bottleneck's first conv in `convs` (decoder convs follow it)
    bool training = true;
    float act_slope = 0.0f;           // 0: ReLU; > 0: LeakyReLU(negative_slope) (UNetDifferentActivation)
    bool compute_bf16 = false;        // conv / wgrad MFMAs on bf16-rounded operands (fp32 storage + accumulate)
    bool compute_x3 = true;           // DEFAULT: float32 contractions by 3 x bf16 pieces (float32-level accuracy)
    bool use_w3() const { return compute_x3 || (compute_bf16 && rfi::bf16_k16()); }   // who reads the pre-split filter records
    bool fuse_bn_bwd = false;         // BatchNorm-backward sums folded into the epilogue of the kernel producing dA (RFI_BN_FUSE=1: on;
                                      // measured neutral: -0.31 ms of BatchNorm passes, +0.18 ms of conv epilogues)
    int loss_kind = 0;                // 0: BCE-with-logits + dice (the reference's, train_model.py:120-128); 1: focal
    float focal_alpha = 0.25f, focal_gamma = 2.0f;
    bool head_sigmoid = false;        // UNetOverfit: forward returns sigmoid(logits); the loss sees that too
    int probs = -1;                   // buffer index of sigmoid(logits) when head_sigmoid

    std::vector<rfi::ConvBN> convs;   // enc1.c1, enc1.c2, ..., encD.c2, bott.c1, bott.c2, decD.c1, decD.c2, ..., dec1.c2
    std::vector<rfi::UpConv> ups;     // decD.up ... dec1.up   (index 0 = deepest)
    size_t head_w_off = 0, head_b_off = 0;
    std::vector<rfi::Entry> entries;
    std::unordered_map<std::string, int> entry_index;

    size_t n_flat = 0;                // floats in each flat buffer (padded)
    int64_t n_params = 0;             // true scalar parameter count
    float *params = nullptr, *grads = nullptr, *adam_m = nullptr, *adam_v = nullptr;
    float* grad_acc = nullptr;        // rfi_model_grad_accumulate: sum of the gradients of several backward passes
    float* chan_pool = nullptr;
    float* wd_pool = nullptr;
    float* w3_pool = nullptr;         // pre-split (3 x bf16) filter records, rebuilt with the dgrad layouts
    rfi::bf16_t* ws_pool = nullptr;   // B-operand-order filters of the wave-specialised conv kernel (ConvBN::ws3f / ws3d)
    void* ws_descs = nullptr;
    int ws_n = 0;
    double ws_bytes = 0;
    std::unordered_map<const float*, const rfi::bf16_t*> ws_by_w;    // float32 filter pointer (ConvArgs::w) -> the same filters for conv_ws / gemm_ws
    int ws_P = 0;                     // planes of the copies in ws_pool: 3 (float32 by 3 x bf16), 1 (bf16 operands), 0 (none built)
    int ws_need() const { return planesP ? 0 : compute_x3 ? 3 : (compute_bf16 && !rfi::bf16_k16()) ? 1 : 0; }
    // the copy of filter `w` for the wave-specialised kernels in the current arithmetic (null: none)
    void ws_set(rfi::ConvArgs& a) const {
        auto it = ws_by_w.find(a.w);
        const rfi::bf16_t* p = it == ws_by_w.end() || ws_P != ws_need() ? nullptr : it->second;
        a.wB3 = ws_P == 3 ? p : nullptr;
        a.wB1 = ws_P == 1 ? p : nullptr;
        if (a.wB3 && x3_skipped.count(a.w)) a.w3 = nullptr;   // (no pre-split records are kept for this layer)
    }
    // the plain U-Net keeps no pre-split (3 x bf16) filter records for the layers the wave-specialised kernels cover: at its
    // shapes they never decline.  The other models (detector backbone on 4 x 4 maps, heads) keep every record up to date, so a
    // declined shape runs the round-2 kernel on valid records instead of a temporary copy (an allocation + a stream
    // synchronisation per launch)
    bool x3_skips_ws_layers = false;
    int ws_n_fwd = 0;                 // the first ws_n_fwd descriptors build the forward-direction copies (sources in `params`)
    double ws_bytes_fwd = 0;
    void refresh_ws_weights(int P, int which = 0);      // which: 0 every copy, 1 the forward direction, 2 the input-gradient direction
    // the input-gradient-direction copies (dgrad layout + their B-operand images) are rebuilt on the SIDE stream while the main
    // stream runs the forward pass; the backward pass waits for wd_ready
    hipEvent_t wd_ready = nullptr;
    bool wd_pending = false;
    std::vector<hipEvent_t> skip_done;      // per encoder level: the skip half of concat[l] is written (side stream, forward pass)
    unsigned skip_pending = 0;              // levels whose skip write the main stream has not waited for yet
    bool wd_side_todo = false;        // the side half of a split rebuild has not been enqueued yet
    void side_rebuild_wd();
    bool x3_reads_wd = false;         // the batched 3 x bf16 record rebuild reads dgrad-layout filters
    void wait_wd();
    // conv-bias gradients of the float32 U-Net path: bn_bwd_apply leaves its per-block partial sums in a per-layer region of
    // dbias_pool; ONE batched launch at the end of the backward pass finishes them all (single-GPU steps: with a gradient
    // exchange the buckets need every gradient of a layer when the layer is done)
    float* dbias_pool = nullptr;
    void* dbias_descs = nullptr;
    int dbias_n = 0, dbias_max_c = 0;
    bool dbias_deferred = false;
    size_t head_rec_off = 0;            // the head's region of dbias_pool (its dw / db partials)
    bool head_fin_deferred = false;
    void* x3_descs = nullptr;         // device table of the batched rebuild
    int x3_n = 0;
    int x3_for_ws_P = -1;             // the ws_P the record list was built for (layers with ws copies are left out)
    int x3_for_shape = -1;            // ... and the prepared H, W (a layer whose maps fall under 8 x 8 keeps its records)
    std::unordered_set<const float*> x3_skipped;      // filters whose records are NOT kept up to date (ws_set clears ConvArgs::w3 for them)
    double x3_bytes = 0;
    int64_t adam_step = 0;
    bool wd_dirty = true;
    bool x3_fresh = false;            // the 3 x bf16 records match the current parameters
    void* relayout_descs = nullptr;   // device table for the batched dgrad-layout rebuild
    int relayout_n = 0;
    int64_t relayout_tiles = 0;
    double relayout_bytes = 0;

    // activations / workspaces for the prepared shape
    int pN = 0, pH = 0, pW = 0;
    std::vector<rfi::DevBuf> bufs;
    // indices into bufs
    std::vector<int> encY1, encY2, concat, pool, decY1, decY2, gA, gB, dconcat, dpool;
    std::vector<int> gAe, gBe;        // float32 U-Net path: the encoder phase's gradient tensors (gA / gB are the decoder's)
    int bottY1 = -1, bottY2 = -1, gBottA = -1, gBottB = -1, logits = -1, dlogits = -1;
    int x_stage = -1, x_stage2 = -1, x_pad = -1, out_stage = -1, ws_red = -1, ws_slab = -1, lab_stage = -1;
    double* d_sums = nullptr;         // [0..3] loss sums, [4] grad sumsq
    float* d_scalars = nullptr;       // [0] loss, [1] grad norm
    float last_loss = 0, last_norm = 0;

    void build();
    void prepare(int n, int h, int w);
    // 3-layer CNN (model_cnn.cpp)
    int cY1 = -1, cY2 = -1, cG1 = -1, cG2 = -1;
    void build_cnn3();
    void prepare_cnn3(int n, int h, int w);
    void forward_cnn3(const float* x_dev, int n, int h, int w);
    void backward_cnn3(const float* x_dev, const uint8_t* labels_dev, int n, int h, int w);
    void reset_channel_state();       // running stats 0/1, BN-less layers: scale 1, shift 0
    float* buf(int i) { return bufs[i].p; }
    int new_buf() { bufs.emplace_back(); return (int)bufs.size() - 1; }

    // ---- fully connected box head (model_mlp.cpp; arch 6): depth FC + ReLU layers in_ch -> feat -> feat, head feat -> out_ch
    void build_mlp();
    void prepare_mlp(int n, int h, int w);
    void forward_mlp(const float* x_dev, int n);
    void backward_mlp(const float* x_dev, int n);

    // ---- ResNet-50-FPN backbone with frozen BatchNorm (model_backbone.cpp; arch 5): feat = base width (64), out_ch = FPN channels
    struct BBlock {
        int stage = 0, stride = 1, cin = 0, width = 0, cout = 0, lvl_in = 2, lvl = 2;      // resolution H >> lvl
        int c1 = -1, c2 = -1, c3 = -1, cd = -1;                                           // convs indices (cd: projection shortcut)
        int Y1 = -1, Y2 = -1, Y3 = -1, Yd = -1, A = -1, xs1 = -1, xsA = -1;               // bufs indices
        float *sc4 = nullptr, *sh4 = nullptr;             // conv1's affine coefficients tiled x 4 (space-to-depth input of a stride-2 conv2)
    };
    std::vector<BBlock> bb;
    int fpn_inner[4] = {-1, -1, -1, -1}, fpn_layer[4] = {-1, -1, -1, -1};
    int bY0 = -1, bP0 = -1, bArg = -1, bdW = -1, bS = -1, bCol = -1, bWp = -1, fP6 = -1, fdP6 = -1;
    int stem_kp() const { return (49 * in_ch + 15) / 16 * 16; }      // K of the K-packed 7x7 stem
    int fL[4] = {-1, -1, -1, -1}, fM[4] = {-1, -1, -1, -1}, fP[4] = {-1, -1, -1, -1}, fdM[4] = {-1, -1, -1, -1}, fdP[4] = {-1, -1, -1, -1};
    float* bb_stem_w3 = nullptr;      // 3 x bf16 records of the K-packed stem filters (rebuilt with them every step)
    int bG[6] = {-1, -1, -1, -1, -1, -1};
    int bT[4][3] = {{-1, -1, -1}, {-1, -1, -1}, {-1, -1, -1}, {-1, -1, -1}};   // dY3 / dY2 / dY1 / dYd of a Bottleneck, by block index mod 3: what the side stream's weight gradients read
    bool frozen_dirty = true;         // frozen BatchNorm buffers changed: scale / shift must be recomputed
    void build_backbone();
    void prepare_backbone(int n, int h, int w);
    void refresh_backbone();
    void forward_backbone(const float* x_dev, int n, int h, int w);
    void backward_backbone(const float* x_dev, int n, int h, int w);

    // ---- per-RoI mask head (model_mask.cpp; arch 3): depth conv3x3+ReLU layers, a transposed conv + ReLU, a 1x1 head.
    // The output map is out_scale (= 2) times the input map in each direction
    int out_scale = 1;
    bool ext_dlogits = false;         // backward from caller-provided dlogits (rfi_model_backward_dlogits)
    std::vector<int> mkY, mkG;
    int mkU = -1, mkGU = -1, mkGx = -1, head_wd = -1, head_w3 = -1, head_wd3 = -1;
    // the 1x1 head as a GEMM on the matrix cores (conv kernels forward / input gradient, weight-gradient kernel) instead of the
    // per-pixel VALU kernels written for one output channel: where it has enough outputs (the RPN head's 5 A = 20)
    bool head_on_mfma() const {
        static const bool off = getenv("RFI_HEAD_VALU") != nullptr;       // A/B runs
        return !off && (arch == 3 || arch == 4) && out_ch >= 8 && out_ch % 4 == 0 && in_ch % 4 == 0 && (compute_x3 || compute_bf16);
    }
    void build_mask();
    void prepare_mask(int n, int h, int w);
    void forward_mask(const float* x_dev, int n, int h, int w);
    void backward_mask(const float* x_dev, const uint8_t* labels_dev, int n, int h, int w);

    // ---- ResNet-18-style encoder (model_resnet.cpp): stem + 4 stages of 2 BasicBlocks
    struct ResBlock {
        int c1 = -1, c2 = -1, cd = -1;            // convs indices: conv1, conv2, projection (-1: identity shortcut)
        int stride = 1, cin = 0, cout = 0, level = 1;
        int Y1 = -1, Y2 = -1, Yd = -1, xs = -1, A = -1;      // bufs indices: raw conv outputs, space-to-depth input, block output
    };
    std::vector<ResBlock> blocks;
    int rs_stemY = -1, rs_a0 = -1, rs_g0 = -1, rs_g1 = -1, rs_dX = -1, rs_dS = -1, rs_dW = -1, rs_dzd = -1;
    int rs_dz[2] = {-1, -1}, rs_dA1[2] = {-1, -1};    // by block parity: the side stream's weight gradients still read the last block's
    float* rs_ones = nullptr;         // [max C] ones / zeros: identity scale / shift for the pooling kernels
    float* rs_zeros = nullptr;
    float* rs_wpool = nullptr;        // derived filters of the stride-2 and 1x1 convs
    void build_resnet();
    void prepare_resnet(int n, int h, int w);
    void refresh_resnet_weights();
    rfi::View forward_resnet_encoder(rfi::View x, int n, int h, int w, bool train_mode);
    void backward_resnet_encoder(const float* x_dev, int n, int h, int w);

    // ---- plane data flow (model_planes.cpp): planesP = 0 off (round-1 kernels on float32 tensors), 1 bf16
    // activations (the bfloat16 compute mode), 3 float32 as three bf16 pieces
    int planesP = 0;
    std::vector<rfi::PlaneBuf> pl;
    std::vector<int> pA1e, pSkip, pPool, pUp, pA1d, pdYa, pdYb, pdYaE, pdYbE, upf;
    int pXin = -1, pA1b = -1, pdYbottA = -1, pdYbottB = -1;
    // bf16 data flow (P = 1, feat % 4 == 0): raw conv outputs that only elementwise kernels read are stored as bfloat16
    // (indices into pl; -1: float32 in bufs): both convs of every encoder, the first of the bottleneck and of every
    // decoder, the last decoder's second.  The four tensors a transposed conv reads stay float32 tensors holding
    // bf16-rounded values, so the arithmetic is uniform: every conv output of this mode is a bfloat16 value
    bool y16_flow = false;
    std::vector<int> yE1, yE2, yD1;
    int yB1 = -1, yD2top = -1;
    // bf16 data flow: the gradient tensors the input-gradient convs write (dA of every first conv, the pooled gradients) are
    // bfloat16 too when the level widths are multiples of 16
    bool g16_flow = false;
    std::vector<int> g16A, g16B, g16pool;       // g16A: dA of the second convs where an elementwise kernel produces it (head, max-pool backward)
    int g16BottB = -1;
    // ResNet-style encoder on the bf16 flow (arch 2, feat % 16 == 0; model_planes.cpp): indices into pl.  Every block owns its
    // raw conv outputs, its activation planes and its dY planes (the side stream's weight gradients read them until side_join)
    struct ResPlanes {
        int Y1 = -1, Y2 = -1, Yd = -1, A1 = -1, A = -1, dY1 = -1, dY2 = -1, dYd = -1;
        rfi::bf16_t* wBcls[4] = {nullptr, nullptr, nullptr, nullptr};     // stride 2: input-gradient filters per parity class
        float* cls = nullptr;                                             // ... and their float32 table (launch_w_s2_classes)
    };
    std::vector<ResPlanes> rpb;
    int rpStemY = -1, rpA0 = -1, rpdY0 = -1, rp_dA1 = -1, rp_dX = -1;
    int rp_dz[2] = {-1, -1};
    float* rs_cls_pool = nullptr;
    bool resnet_planes() const { return arch == 2 && planesP == 1; }
    void forward_resnet_planes(rfi::PlaneSeg& cur, int n, int h, int w, bool train_mode);
    void backward_resnet_planes(int n, int h, int w);
    // transposed convs on the plane kernels (bfloat16 flow, init_features % 32 == 0): the DoubleConv outputs they read are bfloat16
    // (yB2, yD2[l]) and activated once into planes (pUpIn[l]: the forward operand AND the weight gradient's); the decoder's first
    // conv writes its input gradient [up | skip] as bfloat16 (g16cat[l]) and the transposed conv's input gradient is bfloat16
    // too (g16BottA, g16A[l + 1]), with the BatchNorm-backward sums of the layer below in its epilogue
    bool convt_planes = false;
    // the skip half (channels C .. 2 C - 1) of the gradient decoder l's first conv sends into its [up | skip] input
    rfi::YRef skip_grad(int l, int C) {
        if (convt_planes) return rfi::YRef(pl[g16cat[l]].p + C, pl[g16cat[l]].pstride);
        return rfi::YRef(buf(dconcat[l]) + C, (int64_t)2 * C);
    }
    std::vector<int> yD2, pUpIn, g16cat;
    int yB2 = -1, g16BottA = -1;
    rfi::bf16_t* wb_pool = nullptr;
    void* wb_descs = nullptr;
    int wb_n = 0;
    double wb_bytes = 0;
    void set_planes(int P);           // switches the data flow; frees plane tensors of another P
    void prepare_planes(int n, int h, int w);
    int wb_n_fwd = 0;                 // descriptors of the forward-direction filter images (the table's first entries)
    double wb_bytes_fwd = 0;
    void refresh_plane_weights(int which = 0);       // 0 everything; 1 forward-direction images; 2 input-gradient direction (+ class tables)
    void forward_planes(const float* x_dev, int n, int h, int w, bool train_mode);
    void backward_planes(const float* x_dev, const uint8_t* labels_dev, int n, int h, int w);

    // backward-pass overlap: wgrad launches go to the context's side stream (see model.cpp)
    int side_seq = 0;
    int side_bound = 2;               // main may run this many side launches ahead (0: no bound -- nothing the side work reads is rewritten before side_join)
    void side_begin();                // side stream waits for everything enqueued on the main stream so far
    hipEvent_t next_fork_event();     // a fresh event for launch_*(..., done) (null: overlap off)
    void side_begin_after(hipEvent_t producer_done);   // side stream waits for that producer kernel only (null: as side_begin)
    void side_end();                  // marks the side launch; bounds the main stream's run-ahead
    void side_join();                 // main stream waits for all side work
    // a head model inside a larger step (the detector's box / mask heads): nothing the caller does next needs this model's WEIGHT
    // gradients, so the pass ends without waiting for them; whoever touches the gradients, the buffers or the weights next
    // (apply, all-reduce, accumulate, store_grad, the next forward / backward pass) joins first
    hipEvent_t lazy_ev = nullptr;
    bool lazy_pending = false;
    void side_join_lazy();
    void join_pending_side();
    // weight gradients parked for a later point of the backward pass (model.cpp, wgrad_on_side)
    struct DeferredWgrad { rfi::WgradArgs a; hipEvent_t after; int to; };
    std::vector<DeferredWgrad> deferred;
    std::vector<int> defer_to;        // per conv: -2 at once; k >= 0: behind convs[k]'s weight gradient; -1: at the end of the pass
    void wgrad_on_side(int ci, const rfi::WgradArgs& wa, hipEvent_t after, bool after_everything = false);
    void flush_deferred_wgrads();
    // bucketed gradient exchange (common.hpp): grads[lo, hi) are final once everything enqueued so far on the main
    // and side streams has run -> all-reduce them on the communication stream; exchange_join: main waits for all
    bool exchange_in_backward = false;   // set by the full-step entry points only (the split API exchanges explicitly)
    void bucket_ready(size_t lo, size_t hi);
    size_t pend_lo = 0, pend_hi = 0;  // finished but not yet exchanged range (small buckets wait for their neighbours)
    void flush_bucket();
    void exchange_join();
    // BN-apply + activation of layer c as a load transform for its consumers (slope 0 = ReLU)
    rfi::InXform bn_xf(const rfi::ConvBN& c) const { return rfi::act_xform(c.scale(), c.shift(), act_slope); }
    void refresh_dgrad_weights();
    rfi::View network_input(const float* x_dev, int n, int h, int w);
    void forward(const float* x_dev, int n, int h, int w, bool train_mode);
    void loss_forward(const uint8_t* labels_dev, int n, int h, int w);
    void backward(const float* x_dev, const uint8_t* labels_dev, int n, int h, int w);
    void apply(const rfi_hyper& hp, float grad_scale);
    ~rfi_model();
};

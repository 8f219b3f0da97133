/* libl2i_hip.so -- C ABI of the MI355X (gfx950) layout-to-image hot path.
 *
 * The reference (wtliao/layout2img) is pure Python; its intended native boundary is the op-level
 * extension `model.roi_layers._C` that setup.py:19-23,48 would have built (roi_align_forward /
 * roi_align_backward taking tensors + scalars). This header generalises that boundary to every
 * device computation on the path named by BASELINE.json: plain pointers and
 * sizes, no torch types, caller-owned buffers, one HIP stream argument, int return code.
 *
 * Conventions
 *   - All activations are NHWC ("channels last"). "stream" tensors are f32, "operand" tensors
 *     (inputs of MFMA kernels) are of type T selected by `dtype`: 0 = f32, 1 = bf16.
 *   - Channel counts seen by the kernels are padded to multiples of 8 by the caller.
 *   - Kernels never allocate, free or synchronise; outputs (and atomically accumulated outputs,
 *     marked "+=") are pre-allocated / pre-zeroed by the caller.
 *   - Return 0 on success, L2I_ERR_ARG (-1) for a rejected argument, L2I_ERR_LAUNCH (-2) when
 *     the HIP runtime reported a launch error. No exceptions cross the ABI.
 *   - `stream` is a hipStream_t (torch.cuda.current_stream().cuda_stream).
 */
#ifndef L2I_H
#define L2I_H
#ifdef __cplusplus
extern "C" {
#endif

#define L2I_OK 0
#define L2I_ERR_ARG (-1)
#define L2I_ERR_LAUNCH (-2)
#define L2I_F32 0
#define L2I_BF16 1

int l2i_version(void);

/* Implicit-GEMM convolution / linear layer, forward and data-gradient (MFMA).
 * Replaces nn.Conv2d 3x3 / 1x1 and nn.Linear as used at model/resnet_generator_app_v2.py:633-639,
 * 657-670 (nearest x2 upsample fused: up2), model/rcnn_discriminator_app.py:297-344 (avg_pool2d(2)
 * fused: pool2, alpha = 0.25) and their input gradients (same kernel on the dgrad weight pack).
 *   out = alpha * pool?(conv(up?(x), w)) + bias ; zeroed where relu_mask <= 0 ; + res
 * x [B,Hi,Wi,Ci] T; w packed [Npad][Kpad] T (l2i_weights_prepare); (Ho,Wo) = conv-output grid
 * (= 2*(Hi,Wi) if up2); out/res/relu_mask/out_op* are [B,Ho>>pool2,Wo>>pool2,Co]. KH in {1,3}.
 * nimg (optional DEVICE int): only the first *nimg of the B images are live -- workgroups whose rows all belong to
 * later images skip the reduction, and every row of a later image is written as zeros. This is how the ROI heads of the
 * discriminator (model/rcnn_discriminator_app.py:148-166) run over the batch's real ROIs only while the launch keeps
 * the fixed, host-sync-free shape R = b*o: ROIs are compacted to the front (reference order, :145-146, 413-417).
 * stats (optional, [2][Co] f32, ZEROED by the caller, needs `out`, Co % 4 == 0, no `nimg`, and the caller's `scratch` of
 * l2i_conv2d_fwd_dual -- this entry point and l2i_conv2d_fwd_sc have none and refuse `stats`): += per-channel sum and sum of
 * squares of `out` over all pixels -- the batch statistics of a following normalisation (model/norm_module.py:163,
 * sync_batchnorm/batchnorm.py:51-53,77-88), gathered by the epilogue instead of a separate pass over `out`. Round 6: every wave
 * STORES one partial row in the tail of `scratch` and two small launches add the rows in a fixed order -- the statistics are
 * bit-identical from run to run, as F.batch_norm's are on the reference's single device; `ws` is ignored (it was the
 * replicated atomic workspace of rounds 2-5). */
int l2i_conv2d_fwd(const void* x, const void* w, const float* bias, const float* res, const void* relu_mask,
                   float* out, void* out_op, void* out_op_raw, int dtype, int B, int Hi, int Wi, int Ci, int Ho, int Wo,
                   int Co, int KH, int up2, int pool2, int relu_op, int Kpad, float alpha, const int* nimg, float* stats,
                   float* ws, void* stream);

/* The same launch with a residual block's 1x1 shortcut folded in (reference model/resnet_generator_app_v2.py:664-678:
 * `out = conv2(h) + c_sc(upsample(x))`; model/rcnn_discriminator_app.py:317-344: `pool(conv2(h)) + pool(c_sc(x))`):
 *   out = alpha * pool?( conv(x) + conv1x1(sc_x at (y >> sc_up2, x >> sc_up2)) ) + bias + sc_bias
 * sc_x [B, sc_Hi, sc_Wi, sc_Ci] T, sc_w [Npad, sc_Kpad] T (forward pack of the 1x1 weight), sc_bias [Co] or null. The
 * shortcut becomes sc_Ci / 64 more K-steps of the 3x3 tile (conv_sc_tail): its result is neither written nor read back as
 * `res`. `res` and `relu_mask` must be null. sc_out (f32, shape of out, caller-owned, contents undefined afterwards) is
 * where the shortcut goes on launches that cannot fold it (split-K grids, the generic kernel, f32 operands, sc_Ci % 64,
 * L2I_SC_FOLD=0): there the library runs it as a separate 1x1 launch and adds it as the residual -- same result. */
int l2i_conv2d_fwd_sc(const void* x, const void* w, const float* bias, const float* res, const void* relu_mask,
                      float* out, void* out_op, void* out_op_raw, int dtype, int B, int Hi, int Wi, int Ci, int Ho, int Wo,
                      int Co, int KH, int up2, int pool2, int relu_op, int Kpad, float alpha, const int* nimg, float* stats,
                      float* ws, const void* sc_x, const void* sc_w, const float* sc_bias, float* sc_out, int sc_Hi,
                      int sc_Wi, int sc_Ci, int sc_up2, int sc_Kpad, void* stream);

/* DUAL launch of the same convolution: the B images are TWO passes of B/2 images each that share every tensor argument but
 * the weight pack -- the discriminator's D(real) and D(fake) of one optimiser step (train_context_app_v2.py:158,167) are two
 * forward passes over the same layer shapes, but the reference's spectral_norm hook runs a power iteration per pass
 * (model/rcnn_discriminator_app.py:10-15), so each pass has its own W / sigma. Images [0, B/2) are multiplied with w (sc_w),
 * images [B/2, B) with w_b (sc_w_b): one launch with twice the tiles instead of two. `nimg` counts the live leading images of
 * EACH half. w_b == NULL: exactly l2i_conv2d_fwd_sc. Needs B even, no `stats`, and (B/2) * Ho a multiple of the tile's pixel
 * rows ((B/2) * Ho * min(Wo, 16) % 256 == 0 always suffices); L2I_ERR_ARG otherwise (the caller issues two launches).
 * scratch (optional, caller-owned f32, as l2i_conv2d_wgrad's; L2I_WGRAD_SCRATCH_FLOATS suffices): with it 3x3 convolutions on 4x4 maps
 * with >= 256 input channels (model/rcnn_discriminator_app.py:94-96 block6; bf16, no `nimg`, f32 result only) run on the
 * weight-stationary split-K kernel -- every workgroup keeps one 64-channel chunk of the pack for 64 output channels and all
 * pixels in LDS, stores its partial tile there, and a second kernel adds the Ci / 64 partial tiles and applies the epilogue. */
int l2i_conv2d_fwd_dual(const void* x, const void* w, const float* bias, const float* res, const void* relu_mask,
                        float* out, void* out_op, void* out_op_raw, int dtype, int B, int Hi, int Wi, int Ci, int Ho, int Wo,
                        int Co, int KH, int up2, int pool2, int relu_op, int Kpad, float alpha, const int* nimg, float* stats,
                        float* ws, const void* sc_x, const void* sc_w, const float* sc_bias, float* sc_out, int sc_Hi,
                        int sc_Wi, int sc_Ci, int sc_up2, int sc_Kpad, const void* w_b, const void* sc_w_b, float* scratch,
                        long long scratch_floats, void* stream);

/* Data gradient of the first 3x3 convolution of a pre-activation residual block with the data gradient of the block's 1x1 shortcut
 * folded into the same launch (round 5; replaces the shortcut's own data-gradient launch and the residual read of its f32 result):
 *   out = relu_mask > 0 ? alpha * conv3x3(dh, w) : 0   +   sc_alpha * conv1x1(sc_dy at (y >> sc_up2, x >> sc_up2), sc_w)   + res
 * i.e. dx of  x -> relu -> conv1 -> ...  plus dx of  x -> c_sc -> avg_pool(2)?  (reference model/rcnn_discriminator_app.py:326,336-341;
 * sc_up2 = 1, sc_alpha = 0.25 for a down-sampling block). dh [B,H,W,Ci] bf16 (gradient of conv1's result), w the flipped
 * data-gradient pack of conv1 [Npad][Kpad], relu_mask [B,H,W,Co] bf16 (the block input's ReLU operand), sc_dy [B,sc_Hi,sc_Wi,sc_Ci]
 * bf16 (gradient of the block's result), sc_w the data-gradient pack of the 1x1 weight, res optional f32 (another reader's
 * gradient), out f32 and / or out_op_raw bf16 [B,H,W,Co]. The ReLU mask belongs to the 3x3 part only: it is applied to the
 * accumulators before the shortcut's K-steps (conv_mask_first). sc_out (f32, shape of out, contents undefined afterwards): where
 * the shortcut goes on launches that cannot fold (as l2i_conv2d_fwd_sc) -- same result. scratch: as l2i_conv2d_fwd_dual. bf16 only. */
int l2i_conv2d_dgrad_sc(const void* dh, const void* w, const float* res, const void* relu_mask, float* out, void* out_op_raw,
                        int B, int H, int W, int Ci, int Co, int Kpad, float alpha, const int* nimg,
                        const void* sc_dy, const void* sc_w, float* sc_out, int sc_Hi, int sc_Wi, int sc_Ci, int sc_up2, int sc_Kpad,
                        float sc_alpha, float* scratch, long long scratch_floats, void* stream);

/* Per-launch timing of the two MFMA entry points (bench.py's roofline leg). l2i_timing(1): from now on every kernel
 * launched by l2i_conv2d_fwd (class 0) / l2i_conv2d_wgrad (class 1) carries a start / stop HIP event pair attached to
 * its dispatch (hipExtLaunchKernelGGL: the kernel's own begin / end on the stream it runs on); l2i_timing_read
 * synchronises on them and returns the summed duration and the launch count of a class; l2i_timing(0) stops and frees.
 * Not for use under graph capture. */
int l2i_timing(int on);
int l2i_timing_read(int cls, double* total_ms, int* launches);

/* Tuning hook: force one of the forward-kernel tile configurations (see conv_igemm.hip), -1 = built-in heuristic. */
int l2i_set_conv_config(int cfg);

/* Weight gradient of the same convolution: dw[Co][ldw] += alpha * dYfull^T . im2col(x)
 * (autograd of the layers above). dy [B,Ho>>pool2,Wo>>pool2,Co] T; k order (ky,kx,ci).
 * nimg (optional DEVICE int, needs Ho*Wo % 64 == 0): the reduction covers the first *nimg images only.
 * dbias (optional, [Co] f32, +=): the bias gradient alpha * sum_m dYfull[m, co], summed from the dY tiles the kernel
 * stages anyway (no separate pass over dY).
 * scratch (optional, caller-owned, scratch_floats f32 on the device, not shared between streams): when the pixel
 * reduction is split over several workgroups per tile, each STORES its partial tile there and a second kernel adds the
 * splits into dw (plain stores run at 4.5x the rate of f32 atomics); L2I_WGRAD_SCRATCH_FLOATS always suffices. Without
 * it (or when it is too small) the partial tiles are combined by f32 atomics on dw. */
#define L2I_WGRAD_SCRATCH_FLOATS (32LL << 20)
int l2i_conv2d_wgrad(const void* x, const void* dy, float* dw, int dtype, int B, int Hi, int Wi, int Ci, int Ho, int Wo,
                     int Co, int KH, int up2, int pool2, int ldw, float alpha, const int* nimg, float* dbias,
                     float* scratch, long long scratch_floats, void* stream);
/* The same launch with the weight gradient of the block's 1x1 shortcut folded in: conv2 and the shortcut of a residual block
 * receive the SAME dY (model/rcnn_discriminator_app.py:317-344), so dW_sc [Co, sc_ldw] += alpha * dYfull^T . sc_x
 * (sc_x [B, Ho >> sc_up2, Wo >> sc_up2, sc_Ci] T, read at (y >> sc_up2, x >> sc_up2)) becomes ceil(sc_Ci / 128) more column tiles of this launch that stage the same dY
 * steps; sc_dbias (optional) receives the same bias sum as dbias. Launches that cannot carry it (f32 operands, the
 * general-geometry kernel, L2I_SC_WGRAD=0) run the shortcut as a separate launch: same result. */
int l2i_conv2d_wgrad_sc(const void* x, const void* dy, float* dw, int dtype, int B, int Hi, int Wi, int Ci, int Ho, int Wo,
                        int Co, int KH, int up2, int pool2, int ldw, float alpha, const int* nimg, float* dbias,
                        float* scratch, long long scratch_floats, const void* sc_x, float* sc_dw, int sc_Ci, int sc_up2,
                        int sc_ldw, float* sc_dbias, void* stream);
/* DUAL form (see l2i_conv2d_fwd_dual): the weight gradient of images [0, B/2) is added to dw (sc_dw), that of images [B/2, B)
 * to dw_b (sc_dw_b) -- each pass keeps its own accumulator because the spectral-norm backward corrects each with its own
 * u, v, sigma (l2i_weights_backward2). Both halves add their bias gradient to dbias / sc_dbias. `nimg` counts the live images of
 * EACH half. dw_b == NULL: exactly l2i_conv2d_wgrad_sc. Needs B even and (B/2) * Ho * Wo % 64 == 0, else L2I_ERR_ARG.
 * overwrite != 0: the caller guarantees that this launch is the only writer of the dW slices it touches since they were zeroed
 * (the trainer issues one weight-gradient launch per layer application and pass): the result is STORED -- no f32 atomics where an
 * output tile has a single split, no read-modify-write when the splits are reduced. 0: dW += as everywhere else. */
int l2i_conv2d_wgrad_dual(const void* x, const void* dy, float* dw, int dtype, int B, int Hi, int Wi, int Ci, int Ho, int Wo,
                          int Co, int KH, int up2, int pool2, int ldw, float alpha, const int* nimg, float* dbias,
                          float* scratch, long long scratch_floats, const void* sc_x, float* sc_dw, int sc_Ci, int sc_up2,
                          int sc_ldw, float* sc_dbias, float* dw_b, float* sc_dw_b, int overwrite, void* stream);
/* Tuning hook: co-resident workgroups a weight-gradient launch is sized for (0 = derive from the tile: default). */
int l2i_set_wgrad_blocks(int n);
/* Debug aid: co-resident workgroups per CU for conv instantiation `which` with lds_bytes of dynamic LDS;
 * which = 100: the K-split count of the last halo-kernel launch (> 1: stored partial tiles + the reduce kernel that carries the
 * epilogue, < -1: combined by atomics, +-1: not split). */
int l2i_debug_occupancy(int which, int lds_bytes);


/* Weight arena: spectral-norm power iteration (one step, train mode), sigma, and the forward /
 * dgrad packs of every GEMM-shaped weight of a network in three multi-tensor launches.
 * Replaces torch.nn.utils.spectral_norm's pre-forward hook as installed by conv2d()
 * (model/resnet_generator_app_v2.py:681-686, model/rcnn_discriminator_app.py:10-15) and by
 * nn.utils.spectral_norm(nn.Linear/nn.Embedding) (model/norm_module.py:158-159,
 * model/mask_regression.py:64-81, model/rcnn_discriminator_app.py:95,104-109).
 * layers: 20 x int64 per layer row, tables built by layout2img_amd/arena.py; one call per round (a weight the
 * reference applies twice per forward is iterated twice, second round with clear = 0).
 * packed: only elements with co < Co_p, ci < Ci_p are written -- padding rows / K tails must already be zero
 * (allocate the buffer zero-filled once; it may be reused for later passes).
 * Round 6 -- no atomics: W^T u is summed per block of 64 rows into partial rows that a fold launch (tab_tfold) adds in order, and
 * ||W v||^2 per block of 4 R rows into shares the pack / finish launches add in order, both in the caller's transient `scratch`
 * ([npart_floats shares, at the layers' row-19 offsets | partial rows, at the tables' offsets]; launches sharing it must be
 * ordered on one stream): u, v, sigma and every packed weight are bit-identical from run to run, as the CPU reference's are. */
int l2i_weights_prepare(const long long* layers, int n_layers, const int* tab_wtu, int n_wtu, const int* tab_wv, int n_wv,
                        const int* tab_pack, int n_pack, const int* tab_fin, int n_fin, const float* params,
                        float* sn_state, float* pass_uv, long long uv_len, float* norms, void* packed, int dtype,
                        int training, int clear, const int* tab_tfold, int n_tfold, float* scratch, long long scratch_floats,
                        long long npart_floats, void* stream);

/* Backward of the above: grads[w] += (G - <G,Wbar> u v^T) / sigma for every layer (G = dwbar).
 * ws: the all-zero workspace described at l2i_channel_stats (required; left all-zero; n_dot * passes + passes * n_layers floats of it are used).
 * dot_range: (first entry of the FULL dot table, number of entries) per layer row, int32 [n_layers][2]; dot_base: the index in the full table of
 * tab_dot[0] (a launch may cover a sub-range of whole layers: arena.grad_groups). Round 6: every sn_dot block stores its share of <G, W> and
 * a fold launch adds a layer's shares in order -- no float atomics, the correction term is bit-identical from run to run. */
int l2i_weights_backward(const long long* layers, int n_layers, const int* tab_dot, int n_dot, const int* tab_apply,
                         int n_apply, const float* params, const float* dwbar, const float* pass_uv, float* norms,
                         float* grads, float* ws, const int* dot_range, int dot_base, void* stream);
/* The same for TWO passes that share the weights and are flushed together (the discriminator's real and fake forward of one
 * optimiser step, train_context_app_v2.py:158,167): grads += corr0(dwbar0) + corr1(dwbar1) with each pass's own u, v,
 * sigma -- W is read once, the gradient buffer is read and written once. dwbar1 == NULL: one pass (= l2i_weights_backward).
 * overwrite != 0: the caller guarantees that the weights' slices of `grads` are ZERO (first flush after zero_grad): they are
 * written, not read-modify-written (weights applied several times per forward still add atomically onto the zeros). */
int l2i_weights_backward2(const long long* layers, int n_layers, const int* tab_dot, int n_dot, const int* tab_apply,
                          int n_apply, const float* params, const float* dwbar0, const float* pass_uv0, float* norms0,
                          const float* dwbar1, const float* pass_uv1, float* norms1, float* grads, float* ws, int overwrite,
                          const int* dot_range, int dot_base, void* stream);

/* Per-channel sum / sum of squares over rows of x [rows][C] (grouped): sums/sqsums [G][C] +=.
 * Batch statistics of SynchronizedBatchNorm2d (model/sync_batchnorm/batchnorm.py:51-68), of
 * nn.InstanceNorm2d (rows_per_group = H*W) and bias gradients. raw (optional): operand-dtype copy of x written in
 * the same pass (the dY cast of a convolution's backward).
 * scratch (optional, caller-owned f32, 16-byte aligned, scratch_floats floats, contents undefined afterwards; launches sharing it must
 * be ordered on one stream): a group's rows are split over several workgroups only when their partial rows fit there -- they are
 * stored and summed in a fixed order (round 6: bit-identical from run to run), then added to sums / sqsums with one add per address.
 * ws (l2i_norm_mod_bwd_a and the other entry points that take one): all-zero f32 workspace of L2I_WS_FLOATS floats, left all-zero;
 * kernels sharing one workspace must be ordered on one stream. */
#define L2I_WS_FLOATS (32 * 4 * 1024)
int l2i_channel_stats(const float* x, long long rows, int C, long long rows_per_group, float* sums, float* sqsums,
                      void* raw, int dtype, float* scratch, long long scratch_floats, void* stream);

/* Normalise + modulate + ReLU in one pass. mode 0: ISLA (SpatialAdaptiveSynBatchNorm2d.forward,
 * model/norm_module.py:163-186); mode 1: per-channel affine; mode 2: none.
 * x [B][HW][C] f32; statistics [G][C] with G = 1 (stat_stride 0) or B (stat_stride C); mask
 * [B][O][HW] at x's resolution; wproj/bproj [B][O][C] addressed b*pstride_b + o*pstride_o + c. */
int l2i_norm_mod_fwd(const float* x, int B, int HW, int C, const float* sums, const float* sqsums, float count, float eps,
                     int stat_stride, const float* mask, int O, const float* wproj, const float* bproj,
                     long long pstride_b, long long pstride_o, int mode, int relu, void* out_op, float* out_f32, int dtype,
                     float* run_mean, float* run_var, float momentum, void* stream);
/* (run_mean / run_var, optional, batch statistics only: nn.BatchNorm2d's train-mode running-statistics update
 *  r = (1 - momentum) r + momentum * {mean, unbiased var}, done by the same launch.) */

/* Backward, first pass: dxhat = dy*[y>0]*gamma (may alias dy); s1 += sum dxhat, s2 += sum dxhat*xhat;
 * dwproj/dbproj += ; dmask += (all pre-zeroed by the caller). dy_keep: scratch of dy's size, needed when O > 8.
 * part / part_floats (optional, mode 0 with O <= 8): scratch the workgroups park their partial dwproj / dbproj rows in (at
 * most 512 workgroups x 16 x 128 floats = 4 MiB; contents need not be initialised); a second small launch on the same
 * stream (the one that folds ws) adds them into dwproj / dbproj. Without it: atomics from every workgroup.
 * dmask_fresh = 1: dmask is uninitialised memory and receives the result (= instead of +=; it is written with plain stores
 * where one workgroup owns a pixel's mask gradient, cleared by the library first otherwise). */
int l2i_norm_mod_bwd_a(const float* x, const float* dy, int B, int HW, int C, const float* sums, const float* sqsums,
                       float count, float eps, int stat_stride, const float* mask, int O, const float* wproj,
                       const float* bproj, long long pstride_b, long long pstride_o, int mode, int relu, float* dxhat,
                       float* s1, float* s2, float* dwproj, float* dbproj, float* dmask, float* dy_keep, float* ws,
                       float* part, long long part_floats, int dmask_fresh, void* stream);

/* Backward, second pass: dx (+)= invstd * (dxhat - s1/count - xhat*s2/count).
 * dx_op_bf16 (optional, [rows][C] bf16): also receives the bf16 operand copy of the final dx -- what the backward of the
 * convolution that produced x reads as its dY operand (saves a cast pass over dx). */
int l2i_norm_bwd_b(const float* x, const float* dxhat, const float* sums, const float* sqsums, const float* s1,
                   const float* s2, float* dx, long long rows, int C, long long rows_per_group, float count, float eps,
                   int accumulate, void* dx_op_bf16, void* stream);

/* ROIAlign (torchvision.ops.RoIAlign semantics, aligned=False) with the two-scale routing of
 * model/rcnn_discriminator_app.py:98-99,131-145; rows with valid == 0 give zeros. */
int l2i_roi_align_fwd(const float* feat_s, const float* feat_l, const float* rois, const int* valid, float* out, int R,
                      int C, int P, int Hs, int Ws, float scale_s, int Hl, int Wl, float scale_l, float thr, int sampling,
                      void* out_raw_bf16, void* out_relu_bf16, void* stream);
/* (out_raw_bf16 / out_relu_bf16, optional: bf16 copies of out and of relu(out) written by the same launch -- the operands of
 *  the two ROI heads' first convolutions, model/rcnn_discriminator_app.py:148-166) */
int l2i_roi_align_bwd(const float* rois, const int* valid, const float* dout, float* dfeat_s, float* dfeat_l, int R, int C,
                      int P, int Hs, int Ws, float scale_s, int Hl, int Wl, float scale_l, float thr, int sampling, int B, int fresh,
                      void* dfeat_s_bf16, void* dfeat_l_bf16, void* stream);
/* fresh = 0: dfeat_* += (the caller cleared them). fresh = 1 (B = images in the maps): the maps are uninitialised and receive the
 * gradient; for map widths <= 32, P = 8, C % 32 == 0 a gather kernel then writes every pixel once (no atomics, deterministic sum,
 * no clear) and, when given, the bf16 copies of both maps (dfeat_*_bf16: gather form only -- R <= 1024 besides the conditions
 * above -- an error otherwise). */

/* box_attention core (model/resnet_generator_app_v2.py:79-120; geo == NULL gives the VG variant,
 * model/resnet_generator_vg.py:77-122). q, k, v: rows of D floats, `ld` floats apart (D, or the width of a grouped projection
 * result they are column slices of); out [B][O][D]; geo, prob [B][O][O]; keyvalid [B][O]. bwd: dq, dk, dv rows `ldd` apart. */
int l2i_box_attention_fwd(const float* q, const float* k, const float* v, const float* geo, const int* keyvalid, float* out,
                          float* prob, int B, int O, int D, int ld, float scale, void* stream);
int l2i_box_attention_bwd(const float* q, const float* k, const float* v, const float* geo, const float* prob,
                          const float* dout, float* dq, float* dk, float* dv, float* dgeo, int B, int O, int D, int ld, int ldd,
                          float scale, void* stream);

/* Hinge losses with fused backward (train_context_app_v2.py:159-172,180-187).
 * mode 0: mean relu(1-x); 1: mean relu(1+x); 2: -mean x. loss_out += weight*loss; grad = weight*dloss/dx.
 * count_ptr (device, optional): divisor override = global row count under data parallelism. */
int l2i_hinge_fwd_bwd(const float* x, const int* valid, int n, int mode, float weight, const float* count_ptr,
                      float* loss_out, float* grad, void* stream);

/* L1 pixel loss with fused backward (train_context_app_v2.py:143,184). */
int l2i_l1_fwd_bwd(const float* a, const float* b, long long n, float weight, float* loss_out, float* grad, float* part, void* stream);
/* (part: optional scratch of >= 4096 floats, 16-byte aligned: the workgroups' shares of the loss, added in order behind the launch -- no float atomics) */

/* torch.optim.Adam step over one flat buffer (train_context_app_v2.py:121,127,174,189). step_ptr (optional): device
 * int holding the step count t >= 1, read by the kernel instead of `step` -- lets a captured HIP graph of the whole
 * iteration be replayed with the correct bias corrections. */
int l2i_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
                  int step, float grad_scale, const int* step_ptr, void* stream);

/* f32 stream -> T operand copies (raw and/or ReLU'd). */
int l2i_cast_op(const float* x, void* raw, void* act, long long n, int dtype, void* stream);

/* Split operand of the forward-only "bf16x3" precision mode: x [rows][C] f32 (ReLU'd first when relu != 0) -> out3 [rows][3 C]
 * bf16 = [hi | lo | hi], hi = bf16(x), lo = bf16(x - hi). With the weight packs of l2i_weights_prepare(dtype = 3) -- every
 * forward pack [w_hi | w_hi | w_lo] per tap -- l2i_conv2d_fwd over 3 C input channels accumulates x_hi w_hi + x_lo w_hi + x_hi w_lo:
 * the convolution of the reference's nn.Conv2d / nn.Linear (model/resnet_generator_app_v2.py:633-639) to ~2^-16 relative instead
 * of bf16's 2^-8, at MFMA speed / 3 (generator image L_inf vs the reference 8e-5 in emulation, tools/parity/bf16_layer_promotion.py;
 * bar 1e-3). C % 8 == 0. */
int l2i_split_cast(const float* x, void* out3, long long rows, int C, int relu, void* stream);
/* Measurement aid, no reference counterpart: stores the device wall clock (100 MHz ticks) to *slot in stream order. Launched between the
 * phases of a captured iteration it gives the timeline of a graph replay without a profiler (tools/perf/phase_stamps.py). */
int l2i_debug_stamp(long long* slot, void* stream);

/* ReLU backward on f32 streams: out = g*[mask>0] (+ add). */
int l2i_relu_bwd(const float* g, const float* mask, const float* add, float* out, long long n, void* stream);

/* Bilinear resize (align_corners = False) of N planar h x w maps to H x W: F.interpolate(mask, size, mode="bilinear")
 * on the object masks (model/norm_module.py:172-173, model/resnet_generator_app_v2.py:465-470). */
int l2i_resize_bilinear(const float* in, float* out, long long N, int h, int w, int H, int W, void* stream);

/* Appearance head of the discriminator without the (R, C, C) Gram matrices
 * (model/rcnn_discriminator_app.py:148-157; see csrc/misc.hip): x [R][HW][C] pre-ReLU features, w [C] the first half
 * of the head's Linear(2C -> 1) weight. fwd: out[r] += (1/C^2) sum_p (sum_c a)(sum_c a w), a = relu(x); keeps the two
 * per-position sums s, t [R][HW]. bwd: dx [R][HW][C] (written), dw [C] += ; scratch (required, 16-byte aligned, caller-owned, contents undefined
 * afterwards, >= R * ceil(HW / 16) * C floats + the fold's chunk rows): every workgroup stores its share of dw as a row there and a fold launch
 * adds the rows in order (round 6: no float atomics; the forward has one writer per ROI). */
int l2i_gram_head_fwd(const float* x, const float* w, float* out, float* s_keep, float* t_keep, int R, int HW, int C,
                      void* stream);
int l2i_gram_head_bwd(const float* x, const float* w, const float* s_keep, const float* t_keep, const float* g, float* dx,
                      float* dw, float* scratch, long long scratch_floats, int R, int HW, int C, void* dx_op_bf16, void* stream);
/* (dx_op_bf16, optional, here and in l2i_proj_head_bwd: bf16 copy of dx -- the dY operand of the convolution that produced x) */

/* Class-gathered logits of the generator's mask heads (round 5): the heads end in Conv2d(100, 184, 1) and the only reader of the 184-channel
 * result is gather(m, 1, y) (reference model/resnet_generator_app_v2.py:643-651, 465-466), so only the <= 8 channels of the image's own
 * object classes are computed:  lg[b,o,p] = bias[y[b,o]] + sum_c a[b,p,c] w[y[b,o]][c]  (a [B][HH][Cp] f32 NHWC, w [classes][ldw] f32, y [B][O]
 * int64, lg planar [B][O][HH] f32; O <= 8, C <= 128, Cp >= C a multiple of 4). Backward: da[b,p,c] = sum_o gl[b,o,p] w[y_o][c] (every element of
 * da written, pad channels zero); dw[y_o][c] += sum_p gl a; dbias[y_o] += sum_p gl: first per (pixel part, image, object) STORED into tmp
 * ([(l2i_class_logits_bwd_parts(HH) + 1) * B * O][128] f32, contents undefined before and after; the padding class 0 is carried by most images, so direct
 * atomics on dw[0][:] would serialise), then one workgroup per class (`classes` rows of dw) adds its slots' rows into dw / dbias in a fixed order --
 * no atomics anywhere (round 6). C <= 126.
 * l2i_stage_mask_fwd / _bwd take such planar logits with Cp = 0 (their gradient is then `gl`, dlogits may be null). */
int l2i_class_logits_fwd(const float* a, const float* w, const float* bias, const long long* y, float* lg, int B, int O, int HH, int Cp, int C,
                         int ldw, void* stream);
int l2i_class_logits_bwd(const float* a, const float* w, const long long* y, const float* gl, float* da, float* dw, float* dbias, float* tmp,
                         int classes, int B, int O, int HH, int Cp, int C, int ldw, void* stream);
int l2i_class_logits_bwd_parts(int HH);   /* pixel parts per image of the launch above (rows of `tmp` per (image, object) slot) */

/* Stage-mask blend of the generator (model/resnet_generator_app_v2.py:465-470): per object
 *   out = bilinear(bmask, H) * (1 - a) + sigmoid(logits[..., y]) * nearest(boxm, H) * a,  a = sigmoid(alpha[y]).
 * logits [B][H][H][Cp] f32 NHWC (Cp % 4 == 0); bmask, boxm [B][O][S][S] planar f32 with S = f*H, f = 1 or even;
 * y [B][O] int64 class ids < Cp; alpha [Cp]; out [B][O][H][H]; keep [2][B][O][H][H] (sigmoid and resized bmask, for bwd).
 * bwd: g [B][O][H][H] -> dlogits [B][H][H][Cp] and dbmask [B][O][S][S] (both fully written), dalpha [n_alpha] += (the slots' shares go to
 * `share` [B * O], scratch, and one thread per class adds its slots' shares in slot order: no atomics, round 6); gl [B][O][H][H] is scratch. */
int l2i_stage_mask_fwd(const float* logits, const float* bmask, const float* boxm, const float* alpha, const long long* y,
                       float* out, float* keep, int B, int O, int H, int Cp, int S, void* stream);
int l2i_stage_mask_bwd(const float* g, const float* keep, const float* boxm, const float* alpha, const long long* y, float* gl,
                       float* dlogits, float* dbmask, float* dalpha, int B, int O, int H, int Cp, int S, float* share, int n_alpha, void* stream);

/* Projection heads of the discriminator (model/rcnn_discriminator_app.py:127-129 image head, :160-166 object head):
 *   f[r,c] = scale * sum_p relu(x[r,p,c]);  out[r] = sum_c f[r,c] (wl[c] + emb[y[r]][c]) + bias[0]
 * x [R][HW][C] f32 pre-ReLU; wl [C] and emb rows (emb_stride elements apart) of `dtype` (the pass's packed operands);
 * emb / y / bias may be NULL. feat [R][C] is kept for bwd. bwd: dx [R][HW][C] written; dwl [C] +=, demb[y[r]] += (f32 rows
 * demb_stride apart), dbias[0] += -- each may be NULL. */
int l2i_proj_head_fwd(const float* x, const void* wl, const void* emb, int emb_stride, const long long* y, const float* bias,
                      float scale, float* out, float* feat, int R, int HW, int C, int dtype, void* stream);
int l2i_proj_head_bwd(const float* x, const void* wl, const void* emb, int emb_stride, const long long* y, const float* g,
                      const float* feat, float scale, float* dx, float* dwl, float* demb, int demb_stride, float* dbias, int R,
                      int HW, int C, int dtype, void* dx_op_bf16, void* stream);

/* Class-embedding term of the appearance head (model/rcnn_discriminator_app.py:154-157):
 * out[r] = sum_c emb[y[r]][c] w2[c] + bias[0]; bwd: demb[y[r]][c] += g[r] w2[c], dw2[c] += sum_r g[r] emb[y[r]][c], dbias += sum g. */
int l2i_emb_dot_fwd(const void* emb, int emb_stride, const long long* y, const void* w2, const float* bias, float* out, int R,
                    int C, int dtype, void* stream);
int l2i_emb_dot_bwd(const void* emb, int emb_stride, const long long* y, const void* w2, const float* g, float* demb,
                    int demb_stride, float* dw2, float* dbias, int R, int C, int dtype, void* stream);

/* Pyramid-pooling stages of the PSP mask head (model/resnet_generator_app_v2.py:724-752) as fixed sparse linear maps over
 * the H x H pixels of one image (H*H % 128 == 0, H even, <= 128); bins (NB <= 64 in all) are numbered stage after stage,
 * row-major inside a stage. Pixel -> bins direction as per-pixel tap tables, bins -> ... reductions in separable form:
 *   aidx, aw [HW][TA]     (TA = 12 | 16, zero-padded): the AdaptiveAvgPool2d bins a pixel belongs to
 *   uidx, uw [HW][NS][4]  (zero-padded): the bins of stage s a pixel's bilinear (align_corners=True) sample reads
 *   wx [NQ][H], wy [NB][H], xq [NB]: weight of x-bin q (numbered stage after stage, NQ <= 16) at column x; weight of bin k
 *   at row y; the x-bin of bin k -- of the pooling (pool_fwd) resp. the bilinear map (expand_bwd); qoff [NS+1] = first
 *   x-bin of each stage (device array)
 *  pool_fwd:   pooled[b,k,c] = sum_{y,x} wy[k,y] wx[xq[k],x] feats[b,y,x,c]     feats [B][H][H][C] f32, pooled [B][NB][C];
 *              rows: scratch [B][H][NQ][C] f32
 *  pool_bwd:   dfeats[b,p,c] = add[b,p,c] + add2[b,p,c] + cat[b,p,cat_off+c] + sum_t aw[p,t] dpooled[b,aidx[p,t],c]
 *              add, add2 (f32 [B][HW][C]) and cat (rows of cat_w elements of cat_dtype: the gradient of expand_fwd's result, read
 *              in place instead of expand_bwd's dfeats copy) may each be NULL; dfeats_op (optional): bf16 copy of dfeats
 *  expand_fwd: cat[b,p,:] = [sum_t uw[p,s,t] y[b,uidx[p,s,t],:] for s] ++ feats[b,p,:]     y [B][NB][F], cat [B][HW][NS*F+C]
 *              of `dtype`
 *  expand_bwd: g = d cat (dtype) -> dy[b,k,j] = sum_{y,x} wy[k,y] wx[xq[k],x] g[b,y,x,s(k)*F+j], dfeats [B][HW][C] =
 *              g[..., NS*F:] (NULL: not written); rows: scratch [B][H][NQ][F] f32 */
int l2i_psp_pool_fwd(const float* feats, const float* wx, const float* wy, const int* xq, float* pooled, float* rows, int B, int H,
                     int C, int NB, int NQ, void* stream);
int l2i_psp_pool_bwd(const float* dpooled, const int* aidx, const float* aw, int TA, const float* add, const float* add2,
                     const void* cat, int cat_w, int cat_off, int cat_dtype, float* dfeats, void* dfeats_op, int B, int HW, int C,
                     int NB, void* stream);
int l2i_psp_expand_fwd(const float* feats, const float* y, const int* uidx, const float* uw, void* cat, int B, int HW, int C,
                       int F, int NB, int n_stages, int dtype, void* stream);
int l2i_psp_expand_bwd(const void* g, const float* wx, const float* wy, const int* xq, const int* qoff, float* dy, float* dfeats,
                       float* rows, int B, int H, int C, int F, int NB, int NQ, int n_stages, int dtype, void* stream);

/* ---- layout-side glue of the generator (csrc/layout.hip): one launch per reference function instead of chains of
 * elementwise torch ops. All tensors f32 unless noted; `*_op` are optional operand-dtype copies (dtype: 0 f32, 1 bf16). */

/* relu(WGs(BoxRelationalEmbedding(bbox))) (model/resnet_generator_app_v2.py:17-76,175-180): bbox [B][O][4] (xywh read as corner
 * boxes, as the reference does), dim_mat [8] = 1 / 1000^(k/8), wg [64] + wg_bias [1] = the Linear(64, 1); geo [B][O][O].
 * bwd: dwg [64] += , dbias [1] += over the pairs with geo > 0. */
int l2i_box_geometry_fwd(const float* bbox, const float* dim_mat, const float* wg, const float* wg_bias, float* geo, int B, int O,
                         void* stream);
int l2i_box_geometry_bwd(const float* bbox, const float* dim_mat, const float* geo, const float* dgeo, float* dwg, float* dbias,
                         int B, int O, void* stream);

/* sigmoid + masks_to_layout (utils/bilinear.py:137-192: F.grid_sample, bilinear, zeros, align_corners = False, on the
 * box-relative grid) + bbox_mask (model/resnet_generator_app_v2.py:697-721). m: N maps of M x M logits, element stride
 * m_stride (channel 0 of a padded NHWC tensor); bbox [N][4] xywh; lin [H] = torch.linspace(0, 1, H);
 * bmask, boxm (optional) [N][H][H]. bwd: g [N][H][H] -> dm [N][M][M][d_stride] (element 0 the gradient, the rest zeros). */
int l2i_layout_masks_fwd(const float* m, int m_stride, const float* bbox, const float* lin, float* bmask, float* boxm, int N, int M,
                         int H, void* stream);
int l2i_layout_masks_bwd(const float* m, int m_stride, const float* bbox, const float* lin, const float* g, float* dm, int d_stride,
                         int N, int M, int H, void* stream);

/* y[r, :D] = LayerNorm(a'[r] + b[r]) gamma + beta, y[r, D:ldy] = 0 (model/resnet_generator_app_v2.py:199-214). a' = a, or with
 * perm_O > 0 the h = 1 "concat heads" shuffle of each image's (perm_O, D) matrix (:197-198):
 * a'[img, r, c] = a[img, (r D + c) % O, (r D + c) / O]. mean, rstd [rows] are kept for bwd.
 * bwd: da (a's layout, optional), db [rows][ldb] (optional) written; dgamma, dbeta [D] += . */
int l2i_add_layernorm_fwd(const float* a, int lda, const float* b, int ldb, const float* gamma, const float* beta, float eps,
                          float* y, int ldy, void* y_op, int op_dtype, float* mean, float* rstd, int rows, int D, int perm_O,
                          void* stream);
int l2i_add_layernorm_bwd(const float* a, int lda, const float* b, int ldb, const float* gamma, const float* mean, const float* rstd,
                          const float* dy, int ldy, float* da, float* db, float* dgamma, float* dbeta, int rows, int D, int perm_O,
                          float* scratch, long long scratch_floats, void* stream);   /* scratch (optional): the workgroups' dgamma / dbeta rows, added in order */

/* out[r] = [z[r] (Z) | emb[y[r]] (E) | zeros to ld] (model/resnet_generator_app_v2.py:437-441), keyvalid[r] = y[r] != 0
 * (optional). bwd: demb[y[r]] += g[r, Z:Z+E]. */
int l2i_latent_fwd(const float* z, const float* emb, const long long* y, float* out, void* out_op, int op_dtype, int* keyvalid,
                   int rows, int Z, int E, int ld, void* stream);
int l2i_latent_bwd(const float* g, const long long* y, float* demb, int rows, int Z, int E, int ld, void* stream);

/* A Linear's rows [N][C*P] read as .view(N, C, 4, 4) (model/resnet_generator_app_v2.py:453, model/mask_regression.py:87)
 * -> NHWC [N][P][C] (inverse = 0), or the gradient's way back (inverse = 1); out and / or out_op. */
int l2i_fc_to_nhwc(const float* in, float* out, void* out_op, int op_dtype, long long N, int C, int P, int inverse, void* stream);

/* img [B][C][HW] = tanh(pre [B][HW][Cp][:C]) (model/resnet_generator_app_v2.py:497-499);
 * bwd: dpre [B][HW][Cp] = (1 - img^2) g on the first C channels, zeros on the pad (+ operand copy). */
int l2i_tanh_nchw_fwd(const float* pre, float* img, long long B, int C, int Cp, int HW, void* stream);
int l2i_tanh_nchw_bwd(const float* img, const float* g, float* dpre, void* dpre_op, int op_dtype, long long B, int C, int Cp, int HW,
                      void* stream);

/* Pyramid stages of the PSP head (model/resnet_generator_app_v2.py:741-746): per stage s (bins [off_s, off_s + sizes[s]^2) of
 * every image) raw = pooled W_s^T, BatchNorm2d over the stage's B sizes[s]^2 rows (training: batch statistics, running
 * statistics updated with `momentum` and the unbiased variance; else the running statistics), ReLU.
 * pooled [B][NB][C]; W, gamma, beta, rmean, rvar: HOST arrays of S device pointers ([F][C] resp. [F] each: the stage
 * modules' own tensors); raw, y [B][NB][F]; stat [S][2][F] (mean, rstd) kept.
 * bwd: dy -> draw (scratch), dpooled [B][NB][C], dW [S][F][C], dgamma / dbeta [S][F], all written. */
int l2i_psp_stages_fwd(const float* pooled, const float* const* W, const float* const* gamma, const float* const* beta,
                       float* const* rmean, float* const* rvar, float* raw, float* y, float* stat, int B, int NB, int C, int F, int S,
                       const int* sizes, int training, float eps, float momentum, void* stream);
int l2i_psp_stages_bwd(const float* pooled, const float* const* W, const float* const* gamma, const float* const* beta,
                       const float* raw, const float* stat, const float* dy, float* draw, float* dpooled, float* dW, float* dgamma,
                       float* dbeta, int B, int NB, int C, int F, int S, const int* sizes, int training, void* stream);

/* ROI bookkeeping of the discriminator on the device (model/rcnn_discriminator_app.py:131-146,402-417): xywh in [0,1] ->
 * (batch index, x1, y1, x2, y2) * size, rows COMPACTED by a stable sort on 2 [label == 0] + [two_scale and both sides < 64]:
 * real ROIs first (large, then small: the reference's output order), padding rows behind. bbox [R][4], label [R] int64 (R = b*o
 * <= 1024, image of row r = r / o); rois [R][5], y [R] int64, valid [R] int32, count [1] int32 = number of real ROIs. */
int l2i_roi_layout(const float* bbox, const long long* label, float size, int two_scale, int o, int R, float* rois, long long* y,
                   int* valid, int* count, void* stream);

/* img [B][C][H][W] -> x [B][H][W][Cp] (zero pad channels) and optionally its 2x2 average xs [B][H/2][W/2][Cp]
 * (model/rcnn_discriminator_app.py:311-314), f32 + optional operand copies. bwd: dimg = dx[..., :C] + 0.25 dxs (either NULL). */
int l2i_image_nhwc_fwd(const float* img, float* x, void* x_op, float* xs, void* xs_op, int op_dtype, long long B, int C, int Cp, int H,
                       int W, void* stream);
int l2i_image_nhwc_bwd(const float* dx, const float* dxs, float* dimg, long long B, int C, int Cp, int H, int W, void* stream);

/* Adjoint of l2i_resize_bilinear: dx [N][h][w] (written) = the bilinear weights times g [N][H][W], gathered per input pixel. */
int l2i_resize_bilinear_bwd(const float* g, float* dx, long long N, int h, int w, int H, int W, void* stream);

/* nn.Dropout2d on an NHWC stream given uniform draws u [B][C] (model/resnet_generator_app_v2.py:739):
 * out = in * (u >= prob) / (1 - prob); its own backward with in = dy. C % 4 == 0. */
int l2i_channel_dropout(const float* in, const float* u, float* out, long long B, int HW, int C, float prob, void* stream);

/* Mask regressor, between two of its convolutions (reference model/mask_regression.py:64-95): InstanceNorm2d (no affine,
 * biased variance) -> ReLU -> F.interpolate(scale_factor=2, mode="bilinear", align_corners=False) of the per-object maps
 * x [N][S][S][C] f32, S = 4 or 8 -> out [N][2S][2S][C] f32 and (optional) its operand copy out_op (op_dtype 0 f32 / 1 bf16).
 * bwd: dx [N][S][S][C] (+ optional operand copy dx_op) from g = dL/dout [N][2S][2S][C] and x alone (statistics are
 * recomputed in registers). */
int l2i_in_relu_up2_fwd(const float* x, float* out, void* out_op, int op_dtype, long long N, int S, int C, float eps, void* stream);
int l2i_in_relu_up2_bwd(const float* x, const float* g, float* dx, void* dx_op, int op_dtype, long long N, int S, int C, float eps,
                        void* stream);

/* Plain bilinear x2 (align_corners = False) of NHWC maps x [N][S][S][C] f32 -> out [N][2S][2S][C] f32 (+ out_op, the operand-dtype copy the
 * next convolution reads): F.interpolate(x, size, mode="bilinear") between the convolutions of the VG generator's MaskRegressNet
 * (model/mask_regression.py:20-33,42-58). bwd: the adjoint, dx [N][S][S][C] (written; every value has one writer) + its operand copy. C % 4 == 0. */
int l2i_up2_nhwc_fwd(const float* x, float* out, void* out_op, int op_dtype, long long N, int S, int C, void* stream);
int l2i_up2_nhwc_bwd(const float* g, float* dx, void* dx_op, int op_dtype, long long N, int S, int C, void* stream);

#ifdef __cplusplus
}
#endif
#endif

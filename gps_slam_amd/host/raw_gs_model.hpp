// RawGaussianModel (include/raw_gs_model.h:16-258, src/raw_gs_model.cpp) and SLAMGaussianModel::addGaussians
// (slam/slam_gs_model.cpp:5-56) on the C-ABI.
//
// Two routes through one optimise iteration, same kernels underneath:
//   * the reference's call sequence  forward() -> computeLoss() -> loss.backward() -> optimizersStep() ->
//     optimizersZeroGrad()  works unchanged: in grad mode gesForward returns tensors whose grad_fn runs the fused
//     backward chain (raster bwd -> preprocess bwd) and leaves the gradients in the parameters' .grad();
//   * trainStep() is the same iteration as ONE C-ABI call (gps_splat_train_step: 10 launch sites, no host sync, no
//     autograd graph) -- what SLAMPipeline::localOptimize uses.
#pragma once
#include "raw_gs_param.hpp"

class RawGaussianModel {
public:
    RawGaussianModel() = default;
    ~RawGaussianModel() = default;

    void loadConfig(const gpsh::Config& config);  // MODEL section keys of configs/release/*/*.yaml
    void updateSH(int curr_iter = -1);            // raw_gs_model.h:30-36

    // raw_gs_model.h:38-50.  -> {"rgb"[H,W,3], "depth"[H,W,1], "alpha"[H,W,1], "radiis"[N], "means2d"[N,2]}
    TensorDict forward(const Camera& cam, const torch::Tensor& ref_depth = torch::Tensor(),
                       const torch::Tensor& base_color = torch::Tensor());
    TensorDict gesForward(const Camera& cam, const torch::Tensor& ref_depth, const torch::Tensor& base_color);
    TensorDict rawForward(const Camera& cam);  // raw_gs_model.cpp:43-185 (render_method "raw")
    // raw_gs_model.cpp:369-417 with the weights every shipped config uses (L1 only; ssim / depth weights 0):
    // -> {"loss", "l1_loss"}
    TensorDict computeLoss(TensorDict& render_res, const Camera& cam, const gpsh::Config& weight_configs,
                           const torch::Tensor& mask = torch::Tensor());

    // One optimise iteration as a single C-ABI call; the L1 loss accumulates in lossSum().
    // next_cam (optional): the camera of the NEXT trainStep() -- its preprocessing forward then runs in the tail of this step's
    // backward + Adam kernel (gps_splat_step::next_viewmat) and that next call skips its preprocessing launch, provided it comes
    // with exactly that camera and nothing else has used the model in between (any other launch on the step buffers disarms it).
    void trainStep(const Camera& cam, const torch::Tensor& ref_depth, const torch::Tensor& base_color,
                   const torch::Tensor& ref_depth_clamped = torch::Tensor(), const Camera* next_cam = nullptr);
    torch::Tensor lossSum() const { return B_.loss; }

    // Allocate everything an iteration at this image size needs (intermediates, Adam state) now instead of lazily on the
    // first forward / initOptimizers -- keeps one-off hipMalloc + memset time out of the frame loop.
    void reserveWorkspace(int width, int height);
    // Capacity check of the binning buffers, meant for a point where the host synchronises anyway (once per keyframe:
    // SLAMPipeline::removeRedundantGs).  The kernels clamp n_isects / n_groups to the buffer capacities and raise a sticky
    // device flag where the reference would have allocated exact sizes (isect_tiles_no_depth.cu:238-262); this reads it back.
    // Overflow since the last check -> throws (Gaussians were dropped from a render / backward); more than half of a
    // capacity in use -> the buffers are doubled before the next iteration.  Returns {n_isects, n_groups} of the last launch.
    std::pair<int64_t, int64_t> checkBinningCapacity();
    void initOptimizers(int max_iterations = -1, float scene_scale = 1);  // raw_gs_model.cpp:654-675
    void optimizersZeroGrad();
    void optimizersStep();
    void prunePoints(const torch::Tensor& deleteMask);  // raw_gs_model.cpp:635-644 (+ removeFromOptimizer)
    int64_t pruneKeep(const torch::Tensor& keepMask);   // prunePoints(~keepMask); returns the number of Gaussians kept
    void setParamsRequireGrad();

    RawGaussianParams& getGaussianParms() { return opt_gs_params; }
    int getMaxSH() const { return maxSH; }
    torch::Tensor getMeans() { return opt_gs_params.getMeans(); }
    torch::Tensor getScales() { return opt_gs_params.getScales(); }
    torch::Tensor getQuats() { return opt_gs_params.getQuats(); }
    torch::Tensor getFeaturesDc() { return opt_gs_params.getFeaturesDc(); }
    torch::Tensor getFeaturesRest() { return opt_gs_params.getFeaturesRest(); }
    torch::Tensor getOpacities() { return opt_gs_params.getOpacities(); }
    torch::Tensor getRealMeans() { return opt_gs_params.getRealMeans(); }
    torch::Tensor getRealScales() { return opt_gs_params.getRealScales(); }
    torch::Tensor getRealOpacities() { return opt_gs_params.getRealOpacities(); }
    int getGaussianNum() { return opt_gs_params.getGaussianNum(); }
    std::string getRenderMethod() const { return render_method; }
    std::vector<torch::Tensor> grads();  // gradients of the last iteration, reference parameter order
    // Adam state as [:N] views: {exp_avg x 6, exp_avg_sq x 6} in reference parameter order (empty before initOptimizers).
    // UNDEFINED between initOptimizers() and the first step: initOptimizers does not zero the buffers -- step 1 of every route
    // (gps_splat_train_step in all three fuse modes, gps_adam_step) takes m = v = 0 without reading them and writes every live row
    // (tests: test_first_step_after_init_optimizers_does_not_read_the_moments, test_adam_step_one_writes_the_moments_without_
    // reading_them, tests/test_adam_libtorch_gpu.py with NaN-filled buffers).  A caller that steps only SOME tensors at step 1, or
    // starts an optimiser at adam_step != 1, must zero them itself.
    std::vector<torch::Tensor> adamState() {
        std::vector<torch::Tensor> out;
        if (!have_opt_) return out;
        applyPendingPrunes();
        const int64_t N = opt_gs_params.getGaussianNum();
        for (auto* vec : {&adam_m_, &adam_v_}) for (auto& t : *vec) out.push_back(t.slice(0, 0, N));
        return out;
    }

    static torch::Tensor clampRefDepth(const torch::Tensor& ref_depth);  // raw_gs_model.cpp:205-207

    RawGaussianParams opt_gs_params;
    torch::Device device = torch::kCUDA;

    // raw_gs_model.h:283-288 + MODEL section defaults (configs/release/replica/office0.yaml)
    int maxSH = 3, degreesToUse = 3, shDegreeInterval = 0;
    int max_gs_radii = 100, tile_size = 16;
    float eps2d = 0.3f, near_plane = 0.01f, far_plane = 1e10f, radius_clip = 0.0f, delta_depth = 0.1f;
    float maxInitScale = 0.01f, minInitScale = -1.0f, defaultOpacities = 0.5f;
    double means_lr = 1.6e-4, scales_lr = 5e-3, quats_lr = 1e-3, featuresDc_lr = 2.5e-3, featuresRest_lr = 5e-4,
           opacities_lr = 5e-2;
    int64_t isect_capacity = 0;  // 0 -> max(1M, 16 * capacity)
    int64_t binning_overflows = 0;  // how often checkBinningCapacity() found the sticky overflow flag raised (and grew the buffers)
    // trainStep: step featuresRest inside the backward kernel (its gradient never reaches HBM; grads()[4] is then stale).
    // The parameter update is bit-identical either way.
    bool fuse_sh_rest_adam = true;
    // trainStep's binning + backward rasterizer: superblock counting sort + column strips (gps_splat_step::v_rows .. set), or
    // the sorted-key binning + 32-pixel-group kernel of the operator-level entry points
    bool strip_backward = true;
    std::string render_method = "ges";
    bool abs_grad = false;       // raw_gs_model.h:293
    torch::Tensor backgrounds;   // raw_gs_model.h:295 ([1,4] device tensor; undefined = none)

    // implementation detail, public for the autograd node
    struct Buffers {
        torch::Tensor radii, means2d, depths, conics, colors, opacities, records, tiles_per_gauss, flatten_ids,
            group_gs_ids, group_starts, tile_offsets, counts, workspace, render_colors, weight_sum, rgb, depth, loss,
            v_render_colors, v_render_alphas, v_means2d, v_conics, v_colors, v_opacities,
            v_rows, pix2, cls_ids, cls_counts;   // the strip backward's buffers (strip_backward)
    };
    gps_splat_step& stepStruct(int W, int H);
    void bindCamera(gps_splat_step& st, const Camera& cam, const torch::Tensor& ref_depth_clamped,
                    const torch::Tensor& base_color, const torch::Tensor& gt_rgb, bool consumes_prefetch = false);
    Buffers& buffers() { return B_; }
    // Every launch that overwrites the per-launch intermediates (radii, means2d, conics, colors, the tile / group tables, the
    // render) gets a new id.  The grad-mode forward stamps it into its autograd node; its backward reads those intermediates
    // and refuses to run if another launch has replaced them in between.
    int64_t launchId() const { return launch_id_; }
    int64_t nextLaunchId() { return ++launch_id_; }
    // the forward the last trainStep() ran ahead for its next_cam: camera arrays, Gaussian count, image size (viewmat == nullptr: none)
    struct PrefetchKey {
        uint64_t camera = 0;   // Camera::pack_serial() of the camera the forward was run for; 0 = nothing run ahead
        int64_t N = 0;
        int W = 0, H = 0;
        uint64_t version = 0;  // RawGaussianParams::version(): an add / prune between the two steps voids the forward run ahead
        bool operator==(const PrefetchKey& o) const {
            return camera == o.camera && N == o.N && W == o.W && H == o.H && version == o.version;
        }
    };
    PrefetchKey prefetched_;

protected:
    Buffers B_;
    gps_splat_step step_{};
    int64_t step_cap_ = -1;
    int64_t launch_id_ = 0;
    int step_w_ = 0, step_h_ = 0;
    // Adam state: capacity-sized exp_avg / exp_avg_sq / grad buffers in reference parameter order + step count
    std::vector<torch::Tensor> adam_m_, adam_v_, adam_g_;
    double lrs_[RawGaussianParams::NUM] = {0, 0, 0, 0, 0, 0};
    int64_t adam_cap_ = -1;
    int adam_step_ = 0;
    bool have_opt_ = false;
    std::vector<torch::Tensor> leaf_;  // parameter leaves handed to autograd by the last grad-mode forward
    std::vector<torch::Tensor> keep_;  // inputs of the last launch, kept alive until the next one
    struct PendingPrune { torch::Tensor keep; int64_t n_before; };
    std::vector<PendingPrune> pending_prunes_;  // prunePoints' Adam-state compactions not carried out yet
    void applyPendingPrunes();
    torch::Tensor host_count_;         // pinned word the mask compaction reports its count in (addGaussians)
    torch::Tensor host_subset_;        // pinned staging buffer of the sampled subset
};

class SLAMGaussianModel : public RawGaussianModel {
public:
    // slam_gs_model.cpp:5-56: sample new Gaussians where sample_mask is set.  frame_maps: "vertex_map", "normal_map".
    // Returns the number of Gaussians added.  `seed_gen` drives the random subset (host generator: n is host-known).
    int addGaussians(const Camera& cam, const TensorDict& frame_maps, const torch::Tensor& sample_mask,
                     float new_gs_sample_ratio, int frame_num, c10::optional<at::Generator> seed_gen = c10::nullopt);
};

// One C-ABI call per launch-plan SEGMENT instead of one ctypes call per kernel (round 5).
//
// A training step of GSSD++ enqueues ~500 launches on up to a dozen HIP streams; from Python every launch is a ctypes call (argument
// conversion + the call: ~30 us) and every stream fork / join two torch calls -- 17.8 ms of host time per fp32 step, which is what eight ranks
// on one host have to fit beside each other.  gssd_plan_run() walks a flat array of ops built ONCE per plan by the host side
// (gssd/planrun.py): LAUNCH = any entry point of this library whose last parameter is the stream (looked up by name, arguments as 64-bit
// words), WAIT = "stream a waits for everything enqueued on stream b so far" (an event record + hipStreamWaitEvent, the fork / join of the
// plan's branch streams).  Semantics are exactly those of the eager Python loop it replaces -- same launches, same order, same streams -- so
// the gradient-segment hook of the data-parallel reducer (gssd/dist.py) still runs between segments, on the host.
//
// The table below is every `int gssd_*(..., gssd_stream_t stream)` of include/gssd_hip.h (tests/test_host_cpu.py checks the two against
// each other); a thunk unpacks the words into the function's own parameter types.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <tuple>
#include <type_traits>
#include <utility>
#include "common.h"

namespace {

template <class T>
inline T plan_arg(uint64_t w) {
    if constexpr (std::is_pointer_v<T>) {
        return reinterpret_cast<T>(static_cast<uintptr_t>(w));
    } else if constexpr (std::is_same_v<T, float>) {
        const uint32_t b = static_cast<uint32_t>(w);
        float f;
        memcpy(&f, &b, 4);
        return f;
    } else if constexpr (std::is_same_v<T, double>) {
        double d;
        memcpy(&d, &w, 8);
        return d;
    } else {
        static_assert(std::is_integral_v<T>, "plan argument types: pointers, integers, float, double");
        return static_cast<T>(static_cast<int64_t>(w));
    }
}

template <class... P, size_t... I>
inline int plan_call(int (*f)(P...), const uint64_t* a, gssd_stream_t s, std::index_sequence<I...>) {
    using T = std::tuple<P...>;
    static_assert(sizeof...(P) - 1 <= GSSD_PLAN_MAX_ARGS, "GSSD_PLAN_MAX_ARGS");
    static_assert(std::is_same_v<std::tuple_element_t<sizeof...(P) - 1, T>, gssd_stream_t>, "the last parameter is the stream");
    return f(plan_arg<std::tuple_element_t<I, T>>(a[I])..., s);
}

template <class... P>
constexpr int plan_nargs(int (*)(P...)) { return (int)sizeof...(P) - 1; }

struct PlanFn {
    const char* name;
    int nargs;
    int (*thunk)(const uint64_t*, gssd_stream_t);
};

template <auto F>
int plan_thunk(const uint64_t* a, gssd_stream_t s) {
    return plan_call(F, a, s, std::make_index_sequence<plan_nargs(F)>{});
}

#define GSSD_PLAN_FN(f) PlanFn{#f, plan_nargs(&f), &plan_thunk<&f>}
const PlanFn g_plan_fns[] = {
    GSSD_PLAN_FN(gssd_event_record_node),
    GSSD_PLAN_FN(gssd_pack_input_nhwc),
    GSSD_PLAN_FN(gssd_unpack_nhwc_to_nchw),
    GSSD_PLAN_FN(gssd_pack_conv_weight),
    GSSD_PLAN_FN(gssd_pack_conv_weights_batched),
    GSSD_PLAN_FN(gssd_conv2d_nhwc_f32),
    GSSD_PLAN_FN(gssd_conv_x6_pack_weight),
    GSSD_PLAN_FN(gssd_conv_patch_x6_pack_weight),
    GSSD_PLAN_FN(gssd_conv2d_nhwc_bf16),
    GSSD_PLAN_FN(gssd_pack_conv_weight_bf16),
    GSSD_PLAN_FN(gssd_cast_f32_bf16),
    GSSD_PLAN_FN(gssd_cast_bf16_f32),
    GSSD_PLAN_FN(gssd_pack_input_nhwc_bf16),
    GSSD_PLAN_FN(gssd_bn_relu_pool_bf16),
    GSSD_PLAN_FN(gssd_bn_finalize_bf16),
    GSSD_PLAN_FN(gssd_l2norm_bf16),
    GSSD_PLAN_FN(gssd_winograd_weight_f32),
    GSSD_PLAN_FN(gssd_conv2d_wgrad_f32),
    GSSD_PLAN_FN(gssd_conv2d_wgrad_bf16),
    GSSD_PLAN_FN(gssd_unpack_conv_weight_grad),
    GSSD_PLAN_FN(gssd_pack_conv_weight_dgrad),
    GSSD_PLAN_FN(gssd_bn_relu_pool_f32),
    GSSD_PLAN_FN(gssd_bn_finalize_f32),
    GSSD_PLAN_FN(gssd_bn_bwd_reduce_f32),
    GSSD_PLAN_FN(gssd_bn_bwd_finalize_f32),
    GSSD_PLAN_FN(gssd_bn_bwd_apply_f32),
    GSSD_PLAN_FN(gssd_bn_bwd_reduce_mixed),
    GSSD_PLAN_FN(gssd_bn_bwd_apply_mixed),
    GSSD_PLAN_FN(gssd_bn_bwd_apply_masked_f32),
    GSSD_PLAN_FN(gssd_colsum_f32),
    GSSD_PLAN_FN(gssd_cast_f64_f32),
    GSSD_PLAN_FN(gssd_l2norm_bwd_f32),
    GSSD_PLAN_FN(gssd_heads_gather_f32),
    GSSD_PLAN_FN(gssd_upsample_insert_f32),
    GSSD_PLAN_FN(gssd_l2norm_f32),
    GSSD_PLAN_FN(gssd_self_attn_core_f32),
    GSSD_PLAN_FN(gssd_self_attn_core_kv_f32),
    GSSD_PLAN_FN(gssd_sa_pool_kv_f32),
    GSSD_PLAN_FN(gssd_sa_unpool_f32),
    GSSD_PLAN_FN(gssd_self_attn_core_bf16v),
    GSSD_PLAN_FN(gssd_self_attn_core_x6_f32),
    GSSD_PLAN_FN(gssd_self_attn_flash_bwd_bf16),
    GSSD_PLAN_FN(gssd_softmax_rows_f32),
    GSSD_PLAN_FN(gssd_slice_and_cat_f32),
    GSSD_PLAN_FN(gssd_spectral_norm_f32),
    GSSD_PLAN_FN(gssd_dcn_im2col_f32),
    GSSD_PLAN_FN(gssd_dcn_im2col_bf16),
    GSSD_PLAN_FN(gssd_dcn_pack_weight_f32),
    GSSD_PLAN_FN(gssd_dcn_forward_f32),
    GSSD_PLAN_FN(gssd_dcn_streamk_reset),
    GSSD_PLAN_FN(gssd_dcn_pack_weight_x6),
    GSSD_PLAN_FN(gssd_dcn_forward_x6),
    GSSD_PLAN_FN(gssd_dcn_forward_x6_ex),
    GSSD_PLAN_FN(gssd_dcn_pack_weight_bf16),
    GSSD_PLAN_FN(gssd_dcn_forward_bf16),
    GSSD_PLAN_FN(gssd_dcn_col2im_f32),
    GSSD_PLAN_FN(gssd_resize_u8_horizontal),
    GSSD_PLAN_FN(gssd_resize_u8_vertical),
    GSSD_PLAN_FN(gssd_input_finish_f32),
    GSSD_PLAN_FN(gssd_bgemm_f32),
    GSSD_PLAN_FN(gssd_bgemm_ex_f32),
    GSSD_PLAN_FN(gssd_rowdot_f32),
    GSSD_PLAN_FN(gssd_softmax_bwd_rows_f32),
    GSSD_PLAN_FN(gssd_sn_weight_grad_f32),
    GSSD_PLAN_FN(gssd_scaled_transpose_f32),
    GSSD_PLAN_FN(gssd_cast_split_f32_bf16),
    GSSD_PLAN_FN(gssd_cast_rows_f32_bf16),
    GSSD_PLAN_FN(gssd_transpose_cast_f32_bf16),
    GSSD_PLAN_FN(gssd_dot_f32),
    GSSD_PLAN_FN(gssd_axpby_f32),
    GSSD_PLAN_FN(gssd_scale_cast_f64_f32),
    GSSD_PLAN_FN(gssd_sa_sigma_grad_f32),
    GSSD_PLAN_FN(gssd_match_batch),
    GSSD_PLAN_FN(gssd_reduce_max_f32),
    GSSD_PLAN_FN(gssd_hnm_loss),
    GSSD_PLAN_FN(gssd_loss_finalize),
    GSSD_PLAN_FN(gssd_loss_finalize_global),
    GSSD_PLAN_FN(gssd_loss_backward),
    GSSD_PLAN_FN(gssd_detect),
    GSSD_PLAN_FN(gssd_softmax_lastdim_f32),
    GSSD_PLAN_FN(gssd_eval_match),
    GSSD_PLAN_FN(gssd_eval_ap),
    GSSD_PLAN_FN(gssd_heads_reduce_f32),
    GSSD_PLAN_FN(gssd_interp_add_f32),
    GSSD_PLAN_FN(gssd_pixellink_final_f32),
    GSSD_PLAN_FN(gssd_pixellink_loss_f32),
    GSSD_PLAN_FN(gssd_interp_add_bwd_f32),
    GSSD_PLAN_FN(gssd_pixellink_final_bwd_f32),
    GSSD_PLAN_FN(gssd_pixellink_loss_bwd_f32),
    GSSD_PLAN_FN(gssd_pixellink_decode_f32),
};
#undef GSSD_PLAN_FN
constexpr int N_PLAN_FNS = (int)(sizeof(g_plan_fns) / sizeof(g_plan_fns[0]));

// events for the WAIT ops: a ring per device (a wait captures the record made just before it, so a slot can be re-recorded as soon as the
// hipStreamWaitEvent that names it has been enqueued).  Threading contract (gssd_hip.h): gssd_plan_run may be called from several host
// threads; a ring's slot is handed out, recorded and waited on under the ring's mutex, so two threads never share a record / wait pair.  The
// ring is chosen by the device that OWNS the awaited stream (hipStreamGetDevice), not by the caller's current device, and its events are
// created with that device current.
constexpr int N_EVENTS = 64;
struct EventRing {
    hipEvent_t ev[N_EVENTS];
    int next = 0;
    bool ready = false;
    std::mutex mu;
};
EventRing g_rings[32];

// dst waits for everything enqueued on src so far; 0 on success
int record_and_wait(hipStream_t src, hipStream_t dst) {
    int cur = 0, dev = -1;
    (void)hipGetDevice(&cur);
    hipDevice_t sdev;
    if (src && hipStreamGetDevice(src, &sdev) == hipSuccess) dev = (int)sdev;
    if (dev < 0) dev = cur;                               // the null stream: the current device's
    if (dev < 0 || dev >= 32) return 1;
    EventRing& r = g_rings[dev];
    std::lock_guard<std::mutex> lock(r.mu);
    if (!r.ready) {
        if (dev != cur && hipSetDevice(dev) != hipSuccess) return 1;
        bool ok = true;
        for (int i = 0; i < N_EVENTS && ok; ++i) ok = hipEventCreateWithFlags(&r.ev[i], hipEventDisableTiming) == hipSuccess;
        if (dev != cur) (void)hipSetDevice(cur);
        if (!ok) return 1;
        r.ready = true;
    }
    hipEvent_t e = r.ev[r.next];
    r.next = (r.next + 1) % N_EVENTS;
    return (hipEventRecord(e, src) != hipSuccess || hipStreamWaitEvent(dst, e, 0) != hipSuccess) ? 1 : 0;
}

}  // namespace

extern "C" int gssd_plan_fn_count(void) { return N_PLAN_FNS; }

extern "C" const char* gssd_plan_fn_name(int index) { return index >= 0 && index < N_PLAN_FNS ? g_plan_fns[index].name : nullptr; }

extern "C" int gssd_plan_fn_index(const char* name) {
    if (!name) return -1;
    for (int i = 0; i < N_PLAN_FNS; ++i)
        if (!strcmp(g_plan_fns[i].name, name)) return i;
    return -1;
}

extern "C" int gssd_plan_fn_nargs(int index) { return index >= 0 && index < N_PLAN_FNS ? g_plan_fns[index].nargs : -1; }

extern "C" int gssd_plan_op_size(void) { return (int)sizeof(gssd_plan_op); }

extern "C" int gssd_plan_run(const gssd_plan_op* ops, int n_ops, const gssd_stream_t* streams, int n_streams, int* failed_at) {
    GSSD_CHECK_ARG(ops && n_ops >= 0 && streams && n_streams > 0);
    if (failed_at) *failed_at = -1;
    for (int i = 0; i < n_ops; ++i) {
        const gssd_plan_op& op = ops[i];
        int rc = GSSD_OK;
        if (op.stream < 0 || op.stream >= n_streams) {
            gssd_set_error("gssd_plan_run: stream index out of range");
            rc = GSSD_EINVAL;
        } else if (op.kind == GSSD_PLAN_LAUNCH) {
            if (op.fn < 0 || op.fn >= N_PLAN_FNS) {
                gssd_set_error("gssd_plan_run: function index out of range");
                rc = GSSD_EINVAL;
            } else if (op.nargs != g_plan_fns[op.fn].nargs) {
                gssd_set_error("gssd_plan_run: argument count does not match the function's parameters");
                rc = GSSD_EINVAL;
            } else {
                rc = g_plan_fns[op.fn].thunk(op.args, streams[op.stream]);
            }
        } else if (op.kind == GSSD_PLAN_WAIT) {
            if (op.fn < 0 || op.fn >= n_streams) {
                gssd_set_error("gssd_plan_run: awaited stream index out of range");
                rc = GSSD_EINVAL;
            } else if (op.fn != op.stream) {
                if (record_and_wait(as_stream(streams[op.fn]), as_stream(streams[op.stream]))) {
                    gssd_set_error("gssd_plan_run: event record / stream wait failed");
                    rc = GSSD_ELAUNCH;
                }
            }
        } else {
            gssd_set_error("gssd_plan_run: unknown op kind");
            rc = GSSD_EINVAL;
        }
        if (rc != GSSD_OK) {
            if (failed_at) *failed_at = i;
            return rc;
        }
    }
    return GSSD_OK;
}

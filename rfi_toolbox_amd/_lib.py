"""ctypes binding of librfi_hip.so (the C ABI declared in include/rfi_hip.h).

The library is the product: there is no CPU fallback.  Importing this module loads the shared
object (built in-tree by ``python -m rfi_toolbox_amd.build``); a missing library raises
ImportError with the build command, and calling any compute entry point without a GPU raises
RuntimeError from the library's own error string.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RFI_HIP_LIB") or os.path.join(_HERE, "librfi_hip.so")     # (override: A/B runs of two builds)

HOST, DEVICE = 0, 1
C128, C64, F64, F32 = 0, 1, 2, 3
U8, FLOAT32 = 0, 1
IMPL_AUTO, IMPL_DIRECT, IMPL_MFMA, IMPL_MFMA_BF16, IMPL_MFMA_BF16X3, IMPL_PLANES_X3, IMPL_PLANES_BF16, IMPL_WS_X3, IMPL_WS_BF16 = 0, 1, 2, 3, 4, 5, 6, 7, 8


class Hyper(C.Structure):
    _fields_ = [("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double),
                ("weight_decay", C.c_double), ("max_grad_norm", C.c_double)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m rfi_toolbox_amd.build` "
            "(needs hipcc; gfx950 cross-compiles without a GPU). rfi_toolbox_amd has no CPU fallback.")
    return C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)


lib = _load()

_vp, _i, _i64, _f, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t
_pi, _pi64, _pf, _pd = C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(C.c_float), C.POINTER(C.c_double)
_pvp = C.POINTER(C.c_void_p)
_cp = C.c_char_p

_PROTOS = {
    "rfi_abi_version": (_i, []),
    "rfi_last_error": (_cp, []),
    "rfi_device_count": (_i, [_pi]),
    "rfi_ctx_create": (_i, [_i, _pvp]),
    "rfi_ctx_destroy": (_i, [_vp]),
    "rfi_ctx_synchronize": (_i, [_vp]),
    "rfi_ctx_set_overlap": (_i, [_vp, _i]),
    "rfi_ctx_stream": (_i, [_vp, _pvp]),
    "rfi_ctx_device_name": (_i, [_vp, _cp, _sz]),
    "rfi_malloc": (_i, [_vp, _sz, _pvp]),
    "rfi_free": (_i, [_vp, _vp]),
    "rfi_memcpy": (_i, [_vp, _vp, _i, _vp, _i, _sz]),
    "rfi_memset": (_i, [_vp, _vp, _i, _sz]),
    "rfi_timer_start": (_i, [_vp]),
    "rfi_timer_stop": (_i, [_vp, _pf]),
    "rfi_profile_enable": (_i, [_vp, _i]),
    "rfi_profile_reset": (_i, [_vp]),
    "rfi_profile_family_count": (_i, []),
    "rfi_profile_family_name": (_cp, [_i]),
    "rfi_profile_get": (_i, [_vp, _i, _pi64, _pd, _pd, _pd]),
    "rfi_profile_dump": (_i, [_vp, _cp]),
    "rfi_unet_create": (_i, [_vp, _i, _i, _i, _i, _pvp]),
    "rfi_cnn3_create": (_i, [_vp, _i, _i, _i, _pvp]),
    "rfi_unet_resnet_create": (_i, [_vp, _i, _i, _i, _pvp]),
    "rfi_mask_head_create": (_i, [_vp, _i, _i, _i, _pvp]),
    "rfi_model_input_grad": (_i, [_vp, _vp, _i]),
    "rfi_rpn_head_create": (_i, [_vp, _i, _i, _i, _pvp]),
    "rfi_box_head_create": (_i, [_vp, _i, _i, _i, _i, _pvp]),
    "rfi_op_add_inplace": (_i, [_vp, _vp, _vp, _i64]),
    "rfi_op_fastrcnn_loss": (_i, [_vp, _vp, _i64, _i, _vp, _vp, _f, _vp, _pf, _pf]),
    "rfi_resnet50_fpn_create": (_i, [_vp, _i, _i, _i, _pvp]),
    "rfi_backbone_forward": (_i, [_vp, _vp, _i, _i, _i, _i, _pvp, _i]),
    "rfi_backbone_backward": (_i, [_vp, _vp, _i, _i, _i, _i, _pvp, _i]),
    "rfi_model_backward_dlogits": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _i]),
    "rfi_model_grad_accumulate": (_i, [_vp, _i]),
    "rfi_model_set_activation": (_i, [_vp, _f]),
    "rfi_model_set_compute_dtype": (_i, [_vp, _i]),
    "rfi_model_set_head_sigmoid": (_i, [_vp, _i]),
    "rfi_model_set_loss": (_i, [_vp, _i, _f, _f]),
    "rfi_model_destroy": (_i, [_vp]),
    "rfi_model_init": (_i, [_vp, C.c_uint64]),
    "rfi_model_entry_count": (_i, [_vp, _pi]),
    "rfi_model_entry_info": (_i, [_vp, _i, C.POINTER(_cp), _pi, _pi64, _pi, _pi]),
    "rfi_model_load_entry": (_i, [_vp, _cp, _vp, _sz]),
    "rfi_model_store_entry": (_i, [_vp, _cp, _vp, _sz]),
    "rfi_model_param_count": (_i, [_vp, _pi64]),
    "rfi_model_set_training": (_i, [_vp, _i]),
    "rfi_model_forward_nhwc": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i]),
    "rfi_model_forward_nchw": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i]),
    "rfi_train_step": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _i, C.POINTER(Hyper), _pf]),
    "rfi_train_forward_backward": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _i, _pf]),
    "rfi_train_apply": (_i, [_vp, C.POINTER(Hyper), _f, _pf]),
    "rfi_model_loss": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _i, _pf]),
    "rfi_train_step_async": (_i, [_vp, _vp, _vp, _i, _i, _i, C.POINTER(Hyper)]),
    "rfi_model_last_loss": (_i, [_vp, _pf, _pf]),
    "rfi_model_grad_buffer": (_i, [_vp, _pvp, _pi64]),
    "rfi_model_param_buffer": (_i, [_vp, _pvp, _pi64]),
    "rfi_model_store_grad": (_i, [_vp, _cp, _vp, _sz]),
    "rfi_model_store_adam": (_i, [_vp, _cp, _vp, _vp, _sz, _pi64]),
    "rfi_model_load_adam": (_i, [_vp, _cp, _vp, _vp, _sz]),
    "rfi_model_set_adam_step": (_i, [_vp, _i64]),
    "rfi_model_eval_batch": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _i, _f, _pi64, _pi64, _pi64]),
    "rfi_model_algorithmic_flops": (_i, [_vp, _i, _i, _i, _pd, _pd]),
    "rfi_model_debug_tensor": (_i, [_vp, _cp, _vp, _sz, _pi64]),
    "rfi_comm_unique_id": (_i, [_vp]),
    "rfi_comm_init": (_i, [_vp, _vp, _i, _i]),
    "rfi_comm_destroy": (_i, [_vp]),
    "rfi_comm_emulate": (_i, [_vp, _i]),
    "rfi_op_roi_align": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _f, _i, _i, _i, _i, _vp]),
    "rfi_op_mask_targets": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _i, _i, _i, _vp]),
    "rfi_op_anchor_match_batched": (_i, [_vp, _vp, _i64, _i64, _vp, _vp, _i, _i, _vp, _f, _f, _i, _vp, _vp, _vp]),
    "rfi_op_nms_batched": (_i, [_vp, _vp, _vp, _i, _i, _f, _vp]),
    "rfi_op_roi_align_backward_sorted": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _f, _i, _i, _i, _i, _vp]),
    "rfi_op_roi_align_backward": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _f, _i, _i, _i, _i, _vp]),
    "rfi_op_fpn_merge": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "rfi_op_box_decode": (_i, [_vp, _vp, _i64, _vp, _i64, _f, _f, _vp]),
    "rfi_op_nms": (_i, [_vp, _vp, _i, _f, _vp, _pi]),
    "rfi_op_anchor_match": (_i, [_vp, _vp, _i64, _vp, _i, _f, _f, _i, _vp, _vp, _vp]),
    "rfi_op_rpn_loss_dev": (_i, [_vp, _vp, _i64, _i, _vp, _vp, _i64, _f, _vp, _vp, _vp]),
    "rfi_op_rpn_loss_ws_bytes": (_sz, []),
    "rfi_op_rpn_loss_devcount": (_i, [_vp, _vp, _i64, _i, _vp, _vp, _vp, _f, _vp, _vp, _vp]),
    "rfi_op_fastrcnn_loss_dev": (_i, [_vp, _vp, _i64, _i, _vp, _vp, _f, _vp, _vp, _vp]),
    "rfi_op_anchor_match_batched_ws": (_i, [_vp, _vp, _i64, _i64, _vp, _vp, _i, _i, _vp, _f, _f, _i, _vp, _vp, _vp, _vp]),
    "rfi_op_segsort_u64": (_i, [_vp, _vp, _i, _i]),
    "rfi_op_sample_keys": (_i, [_vp, _vp, _i, _i, _vp, C.c_uint64, C.c_uint32, C.c_uint32, _vp, _i]),
    "rfi_op_rpn_sample_apply": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "rfi_op_topk_keys": (_i, [_vp, _vp, _i, _i, _i, _vp, _i]),
    "rfi_op_topk_decode": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _f, _f, _f, _vp, _vp, _vp, _i, _i]),
    "rfi_op_proposals_select": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _vp]),
    "rfi_op_roi_sample": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, C.c_uint64, C.c_uint32, C.c_uint32, _vp, _vp, _vp]),
    "rfi_op_roi_compact": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _f, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp,
                                _vp, _vp, _vp, _vp, _vp]),
    "rfi_op_roi_align_ml": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "rfi_op_roi_align_ml_backward": (_i, [_vp, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _i, _i, _i, _i]),
    "rfi_op_bn_add_relu16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _vp]),
    "rfi_op_relu_mask_sum16": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _i, _vp]),
    "rfi_op_bn_backward16": (_i, [_vp, _vp, _vp, _i64, _i, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp]),
    "rfi_readback_begin": (_i, [_vp, _vp, _sz]),
    "rfi_readback_end": (_i, [_vp, _vp, _sz]),
    "rfi_op_rpn_loss": (_i, [_vp, _vp, _i64, _i, _vp, _vp, _i64, _f, _vp, _pf, _pf]),
    "rfi_op_fpn_merge_backward": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "rfi_comm_allreduce_sum_f32": (_i, [_vp, _vp, _i64]),
    "rfi_model_allreduce_grads": (_i, [_vp]),
    "rfi_preprocess_patches": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _i]),
    "rfi_generate_waterfalls": (_i, [_vp, C.c_uint64, _i, _i, _i, _i, C.c_double, _i, _i, C.c_double, _vp, _vp, _i,
                                     _vp, _i, _vp, _i]),
    "rfi_preprocess_real": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, C.c_double, _vp, _i, _vp, _i]),
    "rfi_mad_flags": (_i, [_vp, _vp, _i, _i, _i, _i, _i, C.c_double, _vp, _i]),
    "rfi_patch_any_flag": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _i, _i, _vp]),
    "rfi_preprocess_gather": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _i, _i, _vp, _i, _vp, _i]),
    "rfi_confusion_counts": (_i, [_vp, _vp, _i, _i, _vp, _i, _i, _i64, _pi64, _pi64, _pi64]),
    "rfi_threshold_logits": (_i, [_vp, _vp, _i64, _f, _vp]),
    "rfi_op_conv3x3": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _vp]),
    "rfi_op_conv1x1": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _vp]),
    "rfi_op_conv_s2": (_i, [_vp, _i, _i, _vp, _i, _i, _i, _i, _vp, _i, _vp]),
    "rfi_op_conv_s2_dgrad": (_i, [_vp, _i, _i, _vp, _i, _i, _i, _i, _vp, _i, _vp]),
    "rfi_op_conv_s2_wgrad": (_i, [_vp, _i, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "rfi_op_conv3x3_dgrad": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _i, _vp]),
    "rfi_op_conv3x3_wgrad": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp]),
    "rfi_op_convt2x2": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp]),
    "rfi_op_convt2x2_dgrad": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _i, _vp]),
    "rfi_op_convt2x2_wgrad": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "rfi_op_bn_stats": (_i, [_vp, _vp, _i64, _i, _vp, _vp]),
    "rfi_op_bn_relu_pool": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "rfi_op_pool_bwd_merge": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "rfi_op_bn_relu_backward": (_i, [_vp, _vp, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
}

for _name, (_res, _args) in _PROTOS.items():
    _fn = getattr(lib, _name)          # AttributeError here == header/library mismatch
    _fn.restype = _res
    _fn.argtypes = _args

EXPORTED = tuple(_PROTOS)


class RfiHipError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        raise RfiHipError(lib.rfi_last_error().decode("utf-8", "replace"))


def device_count() -> int:
    n = C.c_int(0)
    check(lib.rfi_device_count(C.byref(n)))
    return n.value

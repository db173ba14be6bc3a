"""
ctypes binding of libpuzzle_mi355.so (C ABI declared in include/puzzle_mi355.h).

The library is the ONLY compute path of this package: if it cannot be loaded, importing any device-facing
module raises — there is no CPU or torch fallback. Status codes map to exceptions the way the reference's
C extension and ctypes wrappers do (Cuda/Source/Core/Common.h:121-139 -> Python exceptions; ValueError for
layout/dtype/dimension problems, e.g. Cuda/Source/Libs/CuBlas.c:350-354).
"""
import ctypes, os
from ctypes import c_int, c_int32, c_int64, c_uint32, c_uint64, c_size_t, c_float, c_void_p, c_char_p, POINTER, byref

LIBNAME = "libpuzzle_mi355.so"
LIBPATH = os.environ.get("PUZZLE_MI355_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), LIBNAME))


class HipError(RuntimeError):
	pass


class HipMemoryError(HipError, MemoryError):
	pass


class CommError(HipError):
	pass


class ConvDesc(ctypes.Structure):
	_fields_ = [(name, c_int) for name in (
		"n", "c", "h", "w", "k", "r", "s", "stride_h", "stride_w", "pad_h", "pad_w", "dil_h", "dil_w", "groups"
	)]


class PrepackJob(ctypes.Structure):
	_fields_ = [("desc", ConvDesc), ("which", c_int), ("algo", c_int), ("w", c_void_p), ("packed", c_void_p)]


class PoolDesc(ctypes.Structure):
	_fields_ = [(name, c_int) for name in (
		"n", "c", "h", "w", "size_h", "size_w", "stride_h", "stride_w", "pad_h", "pad_w", "mode"
	)]


# Dry run (PUZZLE_MI355_DRYRUN=1, host-logic tests only): the library is loaded for its host-side queries (output shapes,
# workspace sizes, kernel-family resolution) but nothing that needs a device is called — allocations hand out fake
# addresses and every launch is appended to `trace` as (entry point, non-pointer arguments). Lets the whole Python side
# of the backend (lazy buffers, fusion, the executor, the data-parallel reducer) run without a GPU and lets tests compare
# the C-ABI call sequences two callers produce. Never a compute path: results are not computed at all.
DRYRUN = os.environ.get("PUZZLE_MI355_DRYRUN", "0") == "1"
trace = []
# Dry run only: a test harness may take over the device-facing entries (callHook(name, args) -> True when it executed the call).
# The build container's tests use it to run the C ABI on host buffers (a numpy emulation that lives with the tests, not in
# this package), so that the Python glue can be exercised with values where no GPU exists. Never set outside dry-run mode.
callHook = None
# Observers of ISSUED calls: each is called with (name, args) right after a device-facing entry returned. The data-parallel
# arena watcher (grid.ArenaWatcher) uses it to learn that the launch writing a gradient block has actually been queued — a
# write barrier alone only says that somebody asked for the address. Empty (one truth test per call) outside such a run.
issueWatchers = []


def _load():
	if not os.path.exists(LIBPATH):
		raise ImportError(
			"%s not found at %s: build it with `python -c 'import __graft_entry__ as g; g.build()'` or "
			"`make -C puzzlelib_amd/csrc` (hipcc --offload-arch=gfx950)" % (LIBNAME, LIBPATH)
		)
	return ctypes.CDLL(LIBPATH, mode=ctypes.RTLD_GLOBAL)


_lib = _load()
_lib.pz_last_error.restype = c_char_p

P = c_void_p
PP = POINTER(c_void_p)

# name -> argtypes; every entry returns int status (checked by the generated wrapper)
_PROTOS = {
	"pz_init": [c_int],
	"pz_device_count": [POINTER(c_int)],
	"pz_device_name": [c_int, c_char_p, c_int],
	"pz_device_arch": [c_int, c_char_p, c_int],
	"pz_device_sync": [],
	"pz_device_mem_info": [POINTER(c_size_t), POINTER(c_size_t)],
	"pz_device_num_cus": [c_int, POINTER(c_int)],

	"pz_malloc": [PP, c_size_t],
	"pz_free": [P],
	"pz_pool_create": [PP],
	"pz_pool_destroy": [P],
	"pz_pool_alloc": [P, c_size_t, PP],
	"pz_pool_release": [P, P],
	"pz_pool_free_held": [P],
	"pz_pool_stats": [P, POINTER(c_size_t), POINTER(c_size_t), POINTER(c_size_t), POINTER(c_size_t)],
	"pz_pool_oom_events": [POINTER(ctypes.c_long)],
	"pz_pool_driver_allocs": [POINTER(ctypes.c_long), POINTER(ctypes.c_double)],
	"pz_host_alloc_pinned": [PP, c_size_t],
	"pz_host_free_pinned": [P],

	"pz_memcpy_h2d": [P, P, c_size_t, P],
	"pz_memcpy_d2h": [P, P, c_size_t, P],
	"pz_memcpy_d2d": [P, P, c_size_t, P],
	"pz_memcpy_2d": [P, c_size_t, P, c_size_t, c_size_t, c_size_t, P],
	"pz_memset_d32": [P, c_uint32, c_size_t, P],
	"pz_strided_copy": [P, POINTER(c_int64), P, POINTER(c_int64), POINTER(c_int64), c_int, P],

	"pz_stream_create": [PP],
	"pz_stream_create_priority": [PP, c_int],
	"pz_stream_destroy": [P],
	"pz_stream_sync": [P],
	"pz_stream_wait_event": [P, P],
	"pz_event_create": [PP],
	"pz_event_destroy": [P],
	"pz_event_record": [P, P],
	"pz_event_sync": [P],
	"pz_event_query": [P, POINTER(c_int)],
	"pz_event_elapsed_ms": [P, P, POINTER(c_float)],

	"pz_conv2d_out_shape": [POINTER(ConvDesc), POINTER(c_int), POINTER(c_int)],
	"pz_conv2d_workspace_bytes": [POINTER(ConvDesc), c_int, c_int, POINTER(c_size_t)],
	"pz_conv2d_workspace_bytes_pre": [POINTER(ConvDesc), c_int, c_int, POINTER(c_size_t)],
	"pz_conv2d_fwd": [POINTER(ConvDesc), P, P, P, P, c_int, P, c_size_t, P],
	"pz_conv2d_fwd_stats_strips": [POINTER(ConvDesc), c_int, POINTER(c_int)],
	"pz_conv2d_epilogue_supported": [POINTER(ConvDesc), c_int, c_int, POINTER(c_int)],
	"pz_conv2d_fwd_relu": [POINTER(ConvDesc), P, P, P, P, P, c_int, P, c_size_t, P],
	"pz_conv2d_bwd_data_gate": [POINTER(ConvDesc), P, P, P, P, c_int, P, c_size_t, P],
	"pz_conv2d_fwd_stats": [POINTER(ConvDesc), P, P, P, P, P, c_int, P, c_size_t, P],
	"pz_conv2d_bwd_data": [POINTER(ConvDesc), P, P, P, c_int, P, c_size_t, P],
	"pz_conv2d_prepack_bytes": [POINTER(ConvDesc), c_int, c_int, POINTER(c_size_t)],
	"pz_conv2d_prepack": [POINTER(PrepackJob), c_int, P],
	"pz_conv2d_fwd_pre": [POINTER(ConvDesc), P, P, P, P, P, c_int, P, c_size_t, P],
	"pz_conv2d_xbn_supported": [POINTER(ConvDesc), c_int, c_int, POINTER(c_int)],
	"pz_conv2d_fwd_xbn": [POINTER(ConvDesc), P, P, c_int, P, P, P, P, P, c_int, P, c_size_t, P],
	"pz_conv2d_bwd_filter_xbn": [POINTER(ConvDesc), P, P, c_int, P, P, P, P, c_float, c_float, c_int, P, c_size_t, P],
	"pz_conv2d_bwd_data_pre": [POINTER(ConvDesc), P, P, P, c_int, P, c_size_t, P],
	"pz_conv2d_bwd_filter": [POINTER(ConvDesc), P, P, P, P, c_float, c_float, c_int, P, c_size_t, P],

	"pz_conv2d_algo_used": [POINTER(ConvDesc), c_int, c_int, POINTER(c_int)],
	"pz_conv_math_set": [c_int],
	"pz_conv_math_get": [POINTER(c_int)],
	"pz_conv_winograd_tile_set": [c_int],
	"pz_conv_winograd_tile_get": [POINTER(c_int)],
	"pz_conv_profile_enable": [c_int],
	"pz_conv_profile_collect": [POINTER(ctypes.c_double), POINTER(ctypes.c_double), POINTER(ctypes.c_longlong)],

	"pz_gemm": [c_int, c_int, c_int, c_int, c_int, c_float, P, c_int, P, c_int, c_float, P, c_int, P],
	"pz_gemm_workspace_bytes": [c_int, c_int, c_int, POINTER(c_size_t)],
	"pz_gemm_ws": [c_int, c_int, c_int, c_int, c_int, c_float, P, c_int, P, c_int, c_float, P, c_int, P, c_size_t, P],

	"pz_bn_workspace_bytes": [c_int, c_int, c_int, POINTER(c_size_t)],
	"pz_bn_fwd_train": [P, P, c_int, c_int, c_int, P, P, P, P, P, P, c_float, c_float, P, c_size_t, P],
	"pz_bn_fwd_infer": [P, P, c_int, c_int, c_int, P, P, P, P, c_float, P],
	"pz_bn_bwd": [P, P, P, c_int, c_int, c_int, P, P, P, P, P, P, c_size_t, P],
	"pz_bn_fwd_train_act": [P, P, c_int, c_int, c_int, P, P, P, P, P, P, c_float, c_float, c_int, P, c_size_t, P],
	"pz_bn_fwd_train_pre": [P, P, c_int, c_int, c_int, P, P, P, P, P, P, c_float, c_float, c_int, P, c_int, P, c_size_t, P],
	"pz_bn_fwd_train_defer": [c_int, c_int, c_int, P, P, P, P, P, P, c_float, c_float, P, c_int, P, P, c_size_t, P],
	"pz_bn_fwd_train_coef": [P, c_int, c_int, c_int, P, P, P, P, P, P, c_float, c_float, P, c_int, P, P, c_size_t, P],
	"pz_bn_bwd_gate": [P, P, P, c_int, c_int, c_int, P, P, P, P, P, P, P, c_size_t, P],
	"pz_bn_bwd_stats": [P, P, c_int, c_int, c_int, P, P, P],
	"pz_bn_bwd_apply_coef": [P, P, P, c_int, c_int, c_int, P, P],
	"pz_bn_apply_add": [P, P, P, P, P, c_int, c_int, c_int, c_int, P],
	"pz_bn_bwd_coef": [c_int, c_int, c_int, P, P, P, P, P, P, P, c_float, c_float, P, P, P],
	"pz_conv2d_bn_fold_supported": [POINTER(ConvDesc), c_int, POINTER(c_int)],
	"pz_conv2d_bwd_data_bn": [POINTER(ConvDesc), P, P, P, P, P, c_int, P, c_size_t, P],
	"pz_conv2d_bwd_filter_bn": [POINTER(ConvDesc), P, P, P, P, P, c_float, c_float, c_int, P, c_size_t, P],
	"pz_conv2d_bwd_data_bnstats_bytes": [POINTER(ConvDesc), c_int, POINTER(c_size_t)],
	"pz_conv2d_bwd_data_bnstats": [POINTER(ConvDesc), P, P, P, P, P, P, P, P, P, c_int, P, c_size_t, P],
	"pz_bn_bwd_gate_from_partials": [P, P, P, c_int, c_int, c_int, P, P, P, P, P, P, P, P],
	"pz_bn_gate_stats": [P, P, P, P, P, c_int, c_int, c_int, P, P, P, P, P, P, P],
	"pz_bn_gate_stats_up2": [P, P, P, P, P, c_int, c_int, c_int, c_int, P, P, P, P, P, P, P],
	"pz_relu_mask_bytes": [c_int, c_int, c_int, POINTER(c_size_t)],
	"pz_bn_apply_add_mask": [P, P, P, P, P, P, c_int, c_int, c_int, c_int, P],
	"pz_bn_bwd_from_partials": [P, P, P, c_int, c_int, c_int, P, P, P, P, P, P, P, c_float, c_float, P, P],
	"pz_bn_bwd_acc": [P, P, P, c_int, c_int, c_int, P, P, P, P, P, P, c_int, P, P, c_float, c_float, P, c_size_t, P],
	"pz_bn_bwd_act": [P, P, P, c_int, c_int, c_int, P, P, P, P, P, P, c_int, P, c_size_t, P],

	"pz_pool2d_out_shape": [POINTER(PoolDesc), POINTER(c_int), POINTER(c_int)],
	"pz_pool2d_fwd": [POINTER(PoolDesc), P, P, P, P],
	"pz_pool2d_fwd_bn_supported": [POINTER(PoolDesc), POINTER(c_int)],
	"pz_pool2d_fwd_bn": [POINTER(PoolDesc), P, P, c_int, P, P, P],
	"pz_pool2d_bwd": [POINTER(PoolDesc), P, P, P, P, P, P],

	"pz_maskpool2d_fwd": [POINTER(PoolDesc), P, P, P, P],
	"pz_maskpool2d_bwd": [POINTER(PoolDesc), P, P, P, P],
	"pz_maxunpool2d_fwd": [P, P, P, c_size_t, c_size_t, c_size_t, P],
	"pz_maxunpool2d_bwd": [P, P, P, c_size_t, c_size_t, c_size_t, P],
	"pz_lrn_fwd": [P, P, P, c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_int, P],
	"pz_lrn_bwd": [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_int, P],
	"pz_svm_cost": [P, P, c_int, c_int, c_int, c_int, P, P, P],
	"pz_cost_pointwise": [c_int, P, P, P, P, P, P, P, c_size_t, c_int, c_int, c_float, c_float, P],
	"pz_prelu_fwd": [P, P, P, c_int, c_int, c_int, c_int, P],
	"pz_prelu_bwd_data": [P, P, P, P, c_int, c_int, c_int, c_int, P],
	"pz_prelu_bwd_params": [P, P, P, c_int, c_int, c_int, P],
	"pz_reflectpad2d_fwd": [P, P, c_size_t, c_int, c_int, c_int, c_int, c_int, c_int, P],
	"pz_reflectpad2d_bwd": [P, P, c_size_t, c_int, c_int, c_int, c_int, c_int, c_int, P],
	"pz_upsample_fwd": [P, P, c_size_t, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P],
	"pz_upsample_bwd": [P, P, c_size_t, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P],
	"pz_ctc_loss": [P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P],
	"pz_embed_fwd": [P, P, P, c_size_t, c_int, P],
	"pz_embed_bwd_params": [P, P, P, c_float, c_size_t, c_int, P],
	"pz_matvec": [P, P, P, c_int, c_int, c_int, c_int, c_float, c_float, P],
	"pz_argmin_rows": [P, c_int, c_int, P, P],
	"pz_argmin_cols": [P, c_int, c_int, c_int, P, P],

	"pz_softmax_fwd": [P, P, c_int, c_int, c_int, P],
	"pz_softmax_bwd": [P, P, P, c_int, c_int, c_int, P],
	"pz_cross_entropy": [P, P, P, c_int, c_int, c_int, P, P, P, c_size_t, P],

	"pz_reduce_sum_rows": [P, c_int, c_int, P, c_float, c_float, P],
	"pz_reduce_sum_cols": [P, c_int, c_int, c_int, P, c_float, c_float, P],
	"pz_argmax_rows": [P, c_int, c_int, P, P],
	"pz_argmax_cols": [P, c_int, c_int, c_int, P, P],
	"pz_bias_add": [P, P, P, c_int, c_int, c_int, c_int, c_int, P],
	"pz_count_neq_i32": [P, P, c_size_t, P, P],
	"pz_cost_accuracy": [c_int, P, P, c_size_t, P, P],
	"pz_kl_divergence": [P, P, P, c_float, c_size_t, P, P],
	"pz_reduce_minmax_f32": [P, c_size_t, c_int, P, P],
	"pz_reduce_minmax_i32": [P, c_size_t, c_int, P, P],
	"pz_dot": [P, P, c_size_t, P, P],
	"pz_asum": [P, c_size_t, P, P],

	"pz_eltwise": [c_int, c_size_t, PP, c_int, POINTER(c_float), c_int, c_int64, c_int64, c_int64, P],
	"pz_multi_add": [c_int, PP, PP, PP, POINTER(c_float), POINTER(c_float), POINTER(c_uint32), P],
	"pz_cast_i32_f32": [P, P, c_size_t, P],
	"pz_cast_f32_i32": [P, P, c_size_t, P],
	"pz_cast_f32_f16": [P, P, c_size_t, P],
	"pz_cast_f16_f32": [P, P, c_size_t, P],

	"pz_rng_create": [c_uint64, PP],
	"pz_rng_destroy": [P],
	"pz_rng_fill_u32": [P, P, c_size_t, P],
	"pz_rng_fill_uniform": [P, P, c_size_t, P],
	"pz_rng_fill_normal": [P, P, c_size_t, c_float, c_float, P],

	"pz_rtc_compile": [c_char_p, c_char_p, POINTER(c_char_p), c_int, PP, POINTER(c_size_t), c_char_p, c_size_t],
	"pz_rtc_free_code": [P],
	"pz_module_load": [P, PP],
	"pz_module_unload": [P],
	"pz_module_function": [P, c_char_p, PP],
	"pz_function_launch": [P, POINTER(c_uint32), POINTER(c_uint32), c_uint32, P, c_size_t, P],

	"pz_comm_unique_id": [c_char_p],
	"pz_comm_init_rank": [PP, c_int, c_char_p, c_int],
	"pz_comm_destroy": [P],
	"pz_comm_probe": [],
	"pz_comm_info": [P, P, P],
	"pz_comm_async_error": [P],
	"pz_comm_wait_event": [P, P, ctypes.c_double],
	"pz_comm_allreduce_sum_f32": [P, P, P, c_size_t, P],
	"pz_comm_allreduce_sum_f32_ranges": [P, P, POINTER(c_size_t), POINTER(c_size_t), c_int, P],
	"pz_comm_broadcast": [P, P, c_size_t, c_int, P],
}

PZ_OK, PZ_ERR_INVALID, PZ_ERR_HIP, PZ_ERR_NOMEM, PZ_ERR_COMM = 0, 1, 2, 3, 4


def lastError():
	msg = _lib.pz_last_error()
	return msg.decode(errors="replace") if msg else ""


def _raise(status, name):
	msg = "%s: %s" % (name, lastError())

	if status == PZ_ERR_INVALID:
		raise ValueError(msg)
	elif status == PZ_ERR_NOMEM:
		raise HipMemoryError(msg)
	elif status == PZ_ERR_COMM:
		raise CommError(msg)
	else:
		raise HipError(msg)


def _bind(name, argtypes):
	fn = getattr(_lib, name)
	fn.argtypes = argtypes
	fn.restype = c_int

	def call(*args):
		status = fn(*args)
		if status != 0:
			_raise(status, name)
		if issueWatchers:
			for watcher in tuple(issueWatchers):
				watcher(name, args)

	call.__name__ = name
	return call


_HOST_ONLY = {
	"pz_conv2d_out_shape", "pz_conv2d_workspace_bytes", "pz_conv2d_workspace_bytes_pre", "pz_conv2d_prepack_bytes", "pz_conv2d_fwd_stats_strips", "pz_conv2d_epilogue_supported", "pz_conv2d_algo_used",
	"pz_conv2d_bn_fold_supported", "pz_conv2d_bwd_data_bnstats_bytes", "pz_conv2d_xbn_supported", "pz_conv2d_fwd_bn_supported", "pz_bn_workspace_bytes", "pz_relu_mask_bytes",
	"pz_pool2d_out_shape", "pz_pool2d_fwd_bn_supported", "pz_pool_oom_events", "pz_pool_driver_allocs", "pz_gemm_workspace_bytes", "pz_conv_math_set", "pz_conv_math_get",
	"pz_conv_winograd_tile_set", "pz_conv_winograd_tile_get",
	"pz_rtc_compile", "pz_rtc_free_code"                  # (hiprtc compiles for gfx950 without a device)
}
_fake = {"next": 0x7000_0000_0000}


def _fakeHandle(nbytes=256):
	addr = _fake["next"]
	_fake["next"] += (int(nbytes) + 255) // 256 * 256
	return addr


def _store(ref, value):
	ref._obj.value = value


def _dry(name, argtypes):
	"""Stand-in for a device-facing entry point: records the call, fabricates handles / addresses where the caller
	expects one back. Pointer arguments are recorded as 1 / 0 (given / null), everything else by value."""
	def call(*args):
		if callHook is not None and callHook(name, args):
			pass
		elif name == "pz_host_alloc_pinned":
			# HOST memory: the caller writes into it (pipeline.HostStager), so the dry run hands out real memory
			block = ctypes.create_string_buffer(max(int(args[1]), 1))
			_fake.setdefault("pinned", {})[ctypes.addressof(block)] = block
			_store(args[0], ctypes.addressof(block))
		elif name == "pz_host_free_pinned":
			_fake.get("pinned", {}).pop(args[0] if isinstance(args[0], int) else getattr(args[0], "value", None), None)
		elif name == "pz_malloc":
			_store(args[0], _fakeHandle(args[1]))
		elif name == "pz_pool_alloc":
			_store(args[2], _fakeHandle(args[1]))
		elif name in ("pz_pool_create", "pz_stream_create", "pz_stream_create_priority", "pz_event_create"):
			_store(args[0], _fakeHandle())
		elif name == "pz_rng_create":
			_store(args[1], _fakeHandle())
		elif name == "pz_comm_init_rank":
			_store(args[0], _fakeHandle())
			_fake["comm"] = (int(args[1]), int(args[3]))
		elif name == "pz_comm_info":
			_store(args[1], _fake["comm"][0])
			_store(args[2], _fake["comm"][1])
		elif name in ("pz_device_count", ):
			_store(args[0], 1)
		elif name == "pz_device_num_cus":
			_store(args[1], 256)
		elif name in ("pz_device_name", "pz_device_arch"):
			args[1].value = b"dry-run gfx950"
		elif name == "pz_device_mem_info":
			_store(args[0], 288 << 30)
			_store(args[1], 288 << 30)
		elif name == "pz_pool_stats":
			for ref in args[1:]:
				_store(ref, 0)
		elif name == "pz_event_elapsed_ms":
			_store(args[2], 0.0)
		elif name == "pz_event_query":
			_store(args[1], 1)
		elif name == "pz_memcpy_d2h":
			ctypes.memset(args[0], 0, args[2])

		rec = []
		for arg, typ in zip(args, argtypes):
			if typ is P or typ is c_char_p:
				rec.append(0 if arg is None or arg == 0 else 1)
			elif hasattr(arg, "_obj"):
				obj = arg._obj
				rec.append(tuple(getattr(obj, f) for f, _ in obj._fields_) if isinstance(obj, ctypes.Structure) else "out")
			elif isinstance(arg, ctypes.Array):
				rec.append("array")
			elif isinstance(arg, (int, float, bytes)) or arg is None:
				rec.append(arg)
			else:
				rec.append("obj")
		trace.append((name, tuple(rec)))
		if issueWatchers:
			for watcher in tuple(issueWatchers):
				watcher(name, args)

	call.__name__ = name
	return call


for _name, _argtypes in _PROTOS.items():
	if DRYRUN and _name not in _HOST_ONLY:
		globals()[_name] = _dry(_name, _argtypes)
	elif hasattr(_lib, _name):
		globals()[_name] = _bind(_name, _argtypes)
	else:
		raise ImportError("%s does not export %s: rebuild the library (make -C puzzlelib_amd/csrc)" % (LIBPATH, _name))

# The layout of a prepared filter operand and every workspace size depend on two process-wide modes of the library (fp32 /
# split math, Winograd output tile). Whoever changes them — any DnnContext, a test, a tool — goes through these wrappers,
# which advance `modeEpoch`; caches keyed by geometry alone (backend.DnnContext.prepared / convDesc / convGeometry) compare
# their epoch with it and start over.
modeEpoch = 0


def _epochBumping(call):
	def wrapper(*args):
		global modeEpoch
		call(*args)
		modeEpoch += 1
	wrapper.__name__ = call.__name__
	return wrapper


pz_conv_math_set = _epochBumping(pz_conv_math_set)
pz_conv_winograd_tile_set = _epochBumping(pz_conv_winograd_tile_set)

pz_version = _lib.pz_version
pz_version.restype = c_int
_lib.pz_build_id.restype = c_char_p


_lib.pz_build_flags.restype = c_char_p


def buildId():
	"""hash of the sources and compiler flags the loaded library was compiled from (csrc/Makefile: BUILD_ID)"""
	return _lib.pz_build_id().decode()


def buildFlags():
	"""compiler flags of the loaded library (csrc/Makefile: FLAGS + EXTRA)"""
	return _lib.pz_build_flags().decode()


def defaultFlags():
	"""the flags of the shipped build as csrc/Makefile states them (ARCH = gfx950, EXTRA empty), or None without the sources"""
	import re
	path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "Makefile")
	if not os.path.exists(path):
		return None
	text = open(path).read()
	flags = re.search(r"^FLAGS\s*\?=\s*(.*)$", text, re.M).group(1)
	arch = re.search(r"^ARCH\s*\?=\s*(.*)$", text, re.M).group(1).strip()
	return " ".join(flags.replace("$(ARCH)", arch).split())


def requireShippedBuild():
	"""bench.py / smoke(): refuse a variant library (measurement rig, experiment switches) and a library older than the tree"""
	flags, want = buildFlags(), defaultFlags()
	if want is not None and flags != want:
		raise RuntimeError("%s was built with flags %r, the shipped build uses %r: rebuild with `make -C puzzlelib_amd/csrc`" % (LIBPATH, flags, want))
	if "-D" in flags:
		raise RuntimeError("%s is a variant build (%s)" % (LIBPATH, flags))
	if sourceId() not in (None, buildId()):
		raise RuntimeError("%s (build %s) was not built from the sources in the tree (%s)" % (LIBPATH, buildId(), sourceId()))


def sourceId():
	"""the same hash over the sources in this tree, or None when they are not there"""
	import glob, hashlib
	here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
	files = sorted(glob.glob(os.path.join(here, "*.hip")) + glob.glob(os.path.join(here, "*.cpp")) + glob.glob(os.path.join(here, "*.h")))
	header = os.path.join(os.path.dirname(here), "..", "include", "puzzle_mi355.h")
	if not files or not os.path.exists(header):
		return None
	digest = hashlib.sha256()
	for path in files + [header, os.path.join(here, "Makefile")]:
		digest.update(open(path, "rb").read())
	digest.update(defaultFlags().encode())
	return digest.hexdigest()[:16]

COMM_ID_BYTES = 128
MULTI_ADD_MAX = 96
COST_BCE, COST_HINGE, COST_SMOOTH_L1, COST_L1_HINGE = 0, 1, 2, 3

# element-wise op ids (enum pz_eltwise_op)
(
	OP_SIGMOID, OP_SIGMOID_DER, OP_TANH, OP_TANH_DER, OP_RELU, OP_RELU_DER, OP_LEAKY_RELU, OP_LEAKY_RELU_DER,
	OP_ELU, OP_ELU_DER, OP_SOFTPLUS, OP_SOFTPLUS_DER, OP_CLIP, OP_CLIP_DER, OP_GELU, OP_GELU_DER,
	OP_DROPOUT, OP_DROPOUT2D, OP_AXPY, OP_ADD, OP_MUL, OP_LINEAR, OP_ABS, OP_WEIGHT_DECAY, OP_L1_PENALTY, OP_L1_GRAD,
	OP_RBM, OP_ADAM, OP_CLASSIC_MOM_SGD, OP_NESTEROV_MOM_SGD, OP_RMSPROP, OP_ADAGRAD, OP_ADADELTA, OP_RMSPROP_GRAVES,
	OP_SMORMS3, OP_ADD3, OP_IADD, OP_IMUL, OP_ADD3_RELU, OP_ADD3_GATE, OP_COUNT
) = range(41)

CONV_ALGO_AUTO, CONV_ALGO_DIRECT, CONV_ALGO_WINOGRAD, CONV_ALGO_IMPLICIT_GEMM = -1, 1, 3, 5
CONV_FWD, CONV_BWD_DATA, CONV_BWD_FILTER = 0, 1, 2
BN_ACT_NONE, BN_ACT_RELU = 0, 1


def declaredSymbols():
	return sorted(list(_PROTOS.keys()) + ["pz_version", "pz_last_error", "pz_build_id", "pz_build_flags"])

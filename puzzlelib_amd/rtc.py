"""
Run-time compiled kernels of the backend object: `SourceModule`, `ElementwiseKernel`, `ReductionKernel` with the constructor and
call signatures of the reference's (Cuda/SourceModule.py:31-393; Cuda/GPUBackend.py:27-30 hangs them on the backend object;
tests: Cuda/SourceModule.py:432-470 eltwiseTest / reductionTest, which Hip/SourceModule.py:182-189 runs on the HIP backend).

The library's own operators are precompiled HIP (csrc/*.hip). This module serves what a CALLER defines at run time: an
element-wise kernel from a C statement over `i`, a two-stage reduction from a map and a reduce expression, or a whole source
module — compiled for gfx950 by hiprtc (pz_rtc_compile: no hipcc, no on-disk cache, no device needed to compile), loaded with
hipModuleLoadData and launched through pz_function_launch with the arguments packed the way the kernel's parameter list lays them
out. Sources are HIP C++; the kernel templates below are this repository's own (64-bit indices, grid-stride loops, an LDS tree
for the reduction) — not the reference's CUDA strings, whose `__shfl_xor_sync` / `cuda_fp16.h` dialect is the thing replaced.

Argument descriptions are pairs (type, name); a type is anything whose str() is a C type — the reference's
Compiler/Codegen/Types objects (`float_t.const.ptr` -> "const float *") as well as plain strings.
"""
import ctypes, json, struct

import numpy as np

from puzzlelib_amd import lib
from puzzlelib_amd.driver import streamHandle
from puzzlelib_amd.gpuarray import GPUArray

BLOCK = 256

# scalar kernel parameters: C type (qualifiers stripped) -> struct format
SCALARS = {
	"float": "f", "double": "d", "int": "i", "unsigned int": "I", "unsigned": "I", "long long": "q", "unsigned long long": "Q",
	"short": "h", "unsigned short": "H", "signed char": "b", "char": "b", "unsigned char": "B", "int32_t": "i", "uint32_t": "I",
	"int64_t": "q", "uint64_t": "Q", "size_t": "Q", "bool": "?",
}
CTYPE_OF_DTYPE = {
	np.dtype(np.float32): "float", np.dtype(np.float64): "double", np.dtype(np.int32): "int", np.dtype(np.uint32): "unsigned int",
	np.dtype(np.int64): "long long", np.dtype(np.uint64): "unsigned long long", np.dtype(np.int16): "short",
	np.dtype(np.uint16): "unsigned short", np.dtype(np.int8): "signed char", np.dtype(np.uint8): "unsigned char",
}


class RtcError(RuntimeError):
	pass


def ctypeName(T):
	"""the C spelling of an argument type: str() of the reference's type objects, or the string itself"""
	return " ".join(str(T).replace("*", " * ").split())


def isPointer(cname):
	return cname.rstrip().endswith("*")


def scalarFormat(cname):
	base = " ".join(w for w in cname.split() if w not in ("const", "volatile", "restrict", "__restrict__"))
	if base not in SCALARS:
		raise RtcError("cannot pass a kernel parameter of type %r by value" % cname)
	return SCALARS[base]


def declaration(cname, name):
	return "%s __restrict__ %s" % (cname, name) if isPointer(cname) else "%s %s" % (cname, name)


def pack(formats, values):
	"""the kernel-argument buffer: every value at its natural alignment ("P" = a device address)"""
	out = bytearray()
	for fmt, value in zip(formats, values):
		size = struct.calcsize("Q" if fmt == "P" else fmt)
		out += b"\x00" * (-len(out) % size)
		out += struct.pack("Q", int(value)) if fmt == "P" else struct.pack(fmt, value)
	return bytes(out)


class Function:
	"""one kernel of a loaded module: func(*args, block=(x, y, z), grid=(x, y, z)[, shared=bytes, stream=]) — device arrays pass
	their address, numpy scalars their value in their own type (Cuda/Source/Core/Module.c:258-290)"""

	def __init__(self, module, name, handle):
		self.module, self.name, self.handle = module, name, handle

	def __call__(self, *args, block, grid, shared=0, stream=None, formats=None, writes=()):
		values, fmts = [], []
		for idx, arg in enumerate(args):
			if isinstance(arg, GPUArray):
				if not arg.contiguous:
					raise ValueError("gpuarray is not contiguous")
				# (without a parameter list nobody knows which arrays a kernel writes: every array is treated as written)
				values.append(arg.ptr if (formats is None or idx in writes) else arg.rptr)
				fmts.append("P")
			elif formats is not None:
				values.append(arg)
				fmts.append(formats[idx])
			elif isinstance(arg, np.generic):
				values.append(arg.item())
				fmts.append(SCALARS[CTYPE_OF_DTYPE[arg.dtype]])
			else:
				raise TypeError("kernel argument %d: pass device arrays and numpy scalars (got %s)" % (idx, type(arg).__name__))
		buf = pack(fmts, values)
		g = (ctypes.c_uint32 * 3)(*(tuple(int(v) for v in grid) + (1, 1))[:3])
		b = (ctypes.c_uint32 * 3)(*(tuple(int(v) for v in block) + (1, 1))[:3])
		lib.pz_function_launch(self.handle, g, b, int(shared), buf, len(buf), streamHandle(stream))


class SourceModule:
	"""SourceModule(source, options=None, includes=None, externC=False, verbose=True, debug=False, name=None): compiled on first use;
	module.getFunction("kernel") / module.kernel -> Function (Cuda/SourceModule.py:31-101)"""

	def __init__(self, source, options=None, includes=None, externC=False, verbose=True, debug=False, name=None, recipe=None):
		self.source = "extern \"C\"\n{\n%s\n}\n" % source if externC else source
		self.options = list(options) if options is not None else self.getDefaultOptions()
		if includes:
			raise NotImplementedError("SourceModule(includes=...): in-memory headers are not offered; use -I options")
		self.verbose, self.debug, self.name = verbose, debug, name
		self.handle, self.functions, self.log = None, {}, None
		if recipe is not None:
			# (what the kernel templates below were filled with, for tools that have no device to run the code object on)
			self.source = "// pz-rtc: %s\n%s" % (json.dumps(recipe, sort_keys=True), self.source)

	@classmethod
	def getDefaultOptions(cls):
		return []                              # (pz_rtc_compile itself adds --offload-arch=gfx950 -O3 -std=c++17)

	def build(self):
		code, size = ctypes.c_void_p(), ctypes.c_size_t(0)
		log = ctypes.create_string_buffer(1 << 16)
		opts = (ctypes.c_char_p * max(len(self.options), 1))(*[o.encode() for o in self.options])
		name = ("%s.hip" % self.name).encode() if self.name is not None else None
		try:
			lib.pz_rtc_compile(self.source.encode(), name, opts, len(self.options), ctypes.byref(code), ctypes.byref(size), log, len(log))
		except ValueError as e:
			listing = "\n".join("%-4d    %s" % (i + 1, line) for i, line in enumerate(self.source.splitlines()))
			raise RtcError("%s\n%s\nSource:\n%s" % (e, log.value.decode(errors="replace"), listing)) from None
		self.log = log.value.decode(errors="replace") or None
		if self.log is not None and self.verbose:
			print(self.log, flush=True)
		try:
			handle = ctypes.c_void_p()
			lib.pz_module_load(code, ctypes.byref(handle))
			self.handle = handle.value
		finally:
			lib.pz_rtc_free_code(code)

	def getFunction(self, name):
		func = self.functions.get(name)
		if func is None:
			if self.handle is None:
				self.build()
			handle = ctypes.c_void_p()
			lib.pz_module_function(self.handle, name.encode(), ctypes.byref(handle))
			func = self.functions[name] = Function(self, name, handle.value)
		return func

	def __getattr__(self, name):
		if name.startswith("_"):
			raise AttributeError(name)
		return self.getFunction(name)

	def __del__(self):
		try:
			if self.__dict__.get("handle") is not None:
				lib.pz_module_unload(self.handle)
		except Exception:
			pass


class Kernel:
	def __init__(self, arguments, name):
		self.arguments = [(ctypeName(T), argname) for T, argname in arguments]
		self.name, self.module = name, None
		self.formats = ["P" if isPointer(c) else scalarFormat(c) for c, _ in self.arguments]
		# arrays a kernel may write: pointer parameters that are not pointers to const
		self.writes = tuple(i for i, (c, _) in enumerate(self.arguments) if isPointer(c) and "const" not in c.split("*")[0].split())

	def parameters(self):
		return ", ".join(declaration(c, n) for c, n in self.arguments)

	def prepare(self, args):
		if len(args) != len(self.arguments):
			raise TypeError("%s expects %d arguments, got %d" % (self.name, len(self.arguments), len(args)))
		size = next((a.size for a in args if isinstance(a, GPUArray)), None)
		if size is None:
			raise TypeError("%s: no device array among the arguments" % self.name)
		for (cname, argname), arg in zip(self.arguments, args):
			if isPointer(cname) != isinstance(arg, GPUArray):
				raise TypeError("%s: argument %s is declared %r" % (self.name, argname, cname))
		return size


ELTWISE = """
%(preambule)s

extern "C" __global__ void %(name)s(%(params)s, long long size)
{
	for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < size; i += (long long)gridDim.x * blockDim.x)
	{
		%(operation)s;
	}
}

extern "C" __global__ void %(name)s_strided(%(params)s, long long start, long long stop, long long step)
{
	const long long count = (stop - start + step - 1) / step;
	for (long long j = blockIdx.x * (long long)blockDim.x + threadIdx.x; j < count; j += (long long)gridDim.x * blockDim.x)
	{
		const long long i = start + j * step;
		%(operation)s;
	}
}
"""


class ElementwiseKernel(Kernel):
	"""ElementwiseKernel(arguments, operation, name, preambule=""): `operation` is a C statement over the element index `i`;
	kernel(*arrays_and_scalars, slice=None, stream=None) — Cuda/SourceModule.py:143-226"""

	def __init__(self, arguments, operation, name, preambule=""):
		super().__init__(arguments, name)
		self.operation, self.preambule = operation, preambule

	def generateSource(self):
		return ELTWISE % dict(preambule=self.preambule, name=self.name, params=self.parameters(), operation=self.operation)

	def __call__(self, *args, **kwargs):
		size = self.prepare(args)
		if self.module is None:
			self.module = SourceModule(self.generateSource(), name=self.name, recipe={
				"kind": "eltwise", "name": self.name, "args": self.arguments, "operation": self.operation, "preambule": self.preambule})
		slc, stream = kwargs.get("slice"), kwargs.get("stream")
		if slc is None:
			func, count, tail, tailfmt = self.module.getFunction(self.name), size, (size, ), ["q"]
		else:
			start = 0 if slc.start is None else slc.start
			stop = size if slc.stop is None else slc.stop
			step = 1 if slc.step is None else slc.step
			if step < 1:
				raise ValueError("slice step must be positive")
			func, count = self.module.getFunction(self.name + "_strided"), max(0, (stop - start + step - 1) // step)
			tail, tailfmt = (start, stop, step), ["q", "q", "q"]
		if count == 0:
			return
		grid = min((count + BLOCK - 1) // BLOCK, 1 << 20)
		func(*args, *tail, block=(BLOCK, 1, 1), grid=(grid, 1, 1), stream=stream, formats=self.formats + tailfmt, writes=self.writes)


def ElementHalf2Kernel(*args, **kwargs):
	raise NotImplementedError("ElementHalf2Kernel: this backend computes in float32 (fp16 is a storage type only)")


REDUCE = """
typedef %(T)s pz_acc_t;

__device__ __forceinline__ pz_acc_t %(name)s_reduce(pz_acc_t a, pz_acc_t b) { return (%(reduceExpr)s); }

extern "C" __global__ void %(name)s_stage1(%(params)s, pz_acc_t *__restrict__ partials, long long size)
{
	__shared__ pz_acc_t tree[%(block)d];
	pz_acc_t acc = %(neutral)s;
	for (long long i = blockIdx.x * (long long)%(block)d + threadIdx.x; i < size; i += (long long)gridDim.x * %(block)d)
		acc = %(name)s_reduce(acc, (pz_acc_t)(%(mapExpr)s));
	tree[threadIdx.x] = acc;
	__syncthreads();
	for (int h = %(block)d / 2; h > 0; h >>= 1)
	{
		if (threadIdx.x < h) tree[threadIdx.x] = %(name)s_reduce(tree[threadIdx.x], tree[threadIdx.x + h]);
		__syncthreads();
	}
	if (threadIdx.x == 0) partials[blockIdx.x] = tree[0];
}

extern "C" __global__ void %(name)s_stage2(const pz_acc_t *__restrict__ partials, pz_acc_t *__restrict__ out, int count)
{
	__shared__ pz_acc_t tree[%(block)d];
	pz_acc_t acc = %(neutral)s;
	for (int i = threadIdx.x; i < count; i += %(block)d) acc = %(name)s_reduce(acc, partials[i]);
	tree[threadIdx.x] = acc;
	__syncthreads();
	for (int h = %(block)d / 2; h > 0; h >>= 1)
	{
		if (threadIdx.x < h) tree[threadIdx.x] = %(name)s_reduce(tree[threadIdx.x], tree[threadIdx.x + h]);
		__syncthreads();
	}
	if (threadIdx.x == 0) out[0] = tree[0];
}
"""


class ReductionKernel(Kernel):
	"""ReductionKernel(outtype, neutral, reduceExpr, mapExpr, arguments, name): out = reduce over i of mapExpr, reduceExpr over the
	accumulators `a`, `b`; kernel(*arrays_and_scalars, allocator=None) -> 0-d device array of `outtype` — Cuda/SourceModule.py:298-393.
	Deterministic: a fixed tree per workgroup, a fixed number of workgroups for a given size."""

	def __init__(self, outtype, neutral, reduceExpr, mapExpr, arguments, name):
		super().__init__(arguments, name)
		self.outtype = np.dtype(outtype)
		if self.outtype not in CTYPE_OF_DTYPE:
			raise NotImplementedError("ReductionKernel: accumulator type %s" % self.outtype)
		self.neutral, self.reduceExpr, self.mapExpr = neutral, reduceExpr, mapExpr

	def generateSource(self):
		return REDUCE % dict(T=CTYPE_OF_DTYPE[self.outtype], name=self.name, params=self.parameters(), neutral=self.neutral,
							 reduceExpr=self.reduceExpr, mapExpr=self.mapExpr, block=BLOCK)

	def __call__(self, *args, **kwargs):
		size = self.prepare(args)
		allocator, stream = kwargs.get("allocator"), kwargs.get("stream")
		if self.module is None:
			self.module = SourceModule(self.generateSource(), name=self.name, recipe={
				"kind": "reduce", "name": self.name, "args": self.arguments, "T": CTYPE_OF_DTYPE[self.outtype], "neutral": self.neutral,
				"reduceExpr": self.reduceExpr, "mapExpr": self.mapExpr, "block": BLOCK})
		blocks = max(1, min((size + BLOCK - 1) // BLOCK, 4 * BLOCK))
		partials = GPUArray.empty((blocks, ), dtype=self.outtype, allocator=allocator)
		out = GPUArray.empty((), dtype=self.outtype, allocator=allocator)
		self.module.getFunction(self.name + "_stage1")(
			*args, partials, size, block=(BLOCK, 1, 1), grid=(blocks, 1, 1), stream=stream, formats=self.formats + ["P", "q"],
			writes=self.writes + (len(args), ))
		self.module.getFunction(self.name + "_stage2")(
			partials, out, blocks, block=(BLOCK, 1, 1), grid=(1, 1, 1), stream=stream, formats=["P", "P", "i"], writes=(1, ))
		return out

// Run-time compiled kernels: what backend.SourceModule / ElementwiseKernel / ReductionKernel stand on. Replaces Driver.compile
// (NVRTC; Cuda/Source/Core/Driver.c:501-515) and Driver.Module / Function (cuModuleLoadData / cuLaunchKernel,
// Cuda/Source/Core/Module.c:258-290; on the reference's HIP backend: `hipcc --genco` + an on-disk cache, Hip/SourceModule.py:61-99).
// The library's own operators are precompiled (the rest of csrc/); this file serves USER kernels — the reference lets a caller
// define element-wise and reduction kernels from C expressions (Cuda/SourceModule.py:143-393) — through hiprtc for gfx950.
// libhiprtc is loaded lazily with dlopen: nobody who does not compile a kernel at run time needs it.
#include "common.h"

#include <dlfcn.h>
#include <cstdlib>
#include <string>
#include <vector>

namespace {

typedef struct _hiprtcProgram *hiprtcProgram;
typedef int hiprtcResult;

struct Hiprtc {
	void *lib = nullptr;
	hiprtcResult (*CreateProgram)(hiprtcProgram *, const char *, const char *, int, const char *const *, const char *const *) = nullptr;
	hiprtcResult (*CompileProgram)(hiprtcProgram, int, const char *const *) = nullptr;
	hiprtcResult (*GetProgramLogSize)(hiprtcProgram, size_t *) = nullptr;
	hiprtcResult (*GetProgramLog)(hiprtcProgram, char *) = nullptr;
	hiprtcResult (*GetCodeSize)(hiprtcProgram, size_t *) = nullptr;
	hiprtcResult (*GetCode)(hiprtcProgram, char *) = nullptr;
	hiprtcResult (*DestroyProgram)(hiprtcProgram *) = nullptr;
	const char *(*GetErrorString)(hiprtcResult) = nullptr;
};

Hiprtc g_rtc;

int load_hiprtc() {
	if (g_rtc.lib) return PZ_OK;
	const char *names[] = {"libhiprtc.so.7", "libhiprtc.so", "/opt/rocm/lib/libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"};
	void *lib = nullptr;
	for (const char *n : names)
		if ((lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
	if (!lib) {
		pz::set_error("cannot load libhiprtc: %s", dlerror());
		return PZ_ERR_HIP;
	}
#define PZ_SYM(field, name)                                             \
	*(void **)(&g_rtc.field) = dlsym(lib, name);                        \
	if (!g_rtc.field) {                                                 \
		pz::set_error("libhiprtc lacks symbol %s", name);               \
		return PZ_ERR_HIP;                                              \
	}
	PZ_SYM(CreateProgram, "hiprtcCreateProgram")
	PZ_SYM(CompileProgram, "hiprtcCompileProgram")
	PZ_SYM(GetProgramLogSize, "hiprtcGetProgramLogSize")
	PZ_SYM(GetProgramLog, "hiprtcGetProgramLog")
	PZ_SYM(GetCodeSize, "hiprtcGetCodeSize")
	PZ_SYM(GetCode, "hiprtcGetCode")
	PZ_SYM(DestroyProgram, "hiprtcDestroyProgram")
	PZ_SYM(GetErrorString, "hiprtcGetErrorString")
#undef PZ_SYM
	g_rtc.lib = lib;
	return PZ_OK;
}

}  // namespace

struct pz_module {
	hipModule_t mod;
};

extern "C" {

// Compiles HIP C++ `source` for gfx950. On success *code is a malloc'ed code object of *code_bytes bytes (pz_rtc_free_code). The
// compiler's log (warnings, or the errors of a failed compilation) is copied into `log` (NUL-terminated, truncated to log_bytes).
// A compilation error is PZ_ERR_INVALID with the first line of the log as the message — no device is needed up to here.
int pz_rtc_compile(const char *source, const char *name, const char *const *options, int noptions, void **code, size_t *code_bytes,
                   char *log, size_t log_bytes) {
	PZ_REQUIRE(source && code && code_bytes, "pz_rtc_compile: null argument");
	if (int rc = load_hiprtc()) return rc;
	if (log && log_bytes) log[0] = 0;
	*code = nullptr, *code_bytes = 0;

	hiprtcProgram prog = nullptr;
	hiprtcResult r = g_rtc.CreateProgram(&prog, source, name ? name : "kernel.hip", 0, nullptr, nullptr);
	if (r != 0) {
		pz::set_error("hiprtcCreateProgram: %s", g_rtc.GetErrorString(r));
		return PZ_ERR_HIP;
	}
	std::vector<const char *> opts{"--offload-arch=gfx950", "-O3", "-std=c++17"};
	for (int i = 0; i < noptions; ++i) opts.push_back(options[i]);
	r = g_rtc.CompileProgram(prog, (int)opts.size(), opts.data());

	size_t ls = 0;
	g_rtc.GetProgramLogSize(prog, &ls);
	std::string text(ls + 1, '\0');
	if (ls > 1) g_rtc.GetProgramLog(prog, &text[0]);
	if (log && log_bytes) {
		strncpy(log, text.c_str(), log_bytes - 1);
		log[log_bytes - 1] = 0;
	}
	if (r != 0) {
		const size_t eol = text.find('\n');
		pz::set_error("compilation of %s failed (%s): %s", name ? name : "a run-time kernel", g_rtc.GetErrorString(r),
		              text.substr(0, eol == std::string::npos ? text.size() : eol).c_str());
		g_rtc.DestroyProgram(&prog);
		return PZ_ERR_INVALID;
	}

	size_t cs = 0;
	g_rtc.GetCodeSize(prog, &cs);
	void *buf = malloc(cs ? cs : 1);
	if (!buf) {
		g_rtc.DestroyProgram(&prog);
		pz::set_error("pz_rtc_compile: out of host memory");
		return PZ_ERR_NOMEM;
	}
	g_rtc.GetCode(prog, (char *)buf);
	g_rtc.DestroyProgram(&prog);
	*code = buf, *code_bytes = cs;
	return PZ_OK;
}

int pz_rtc_free_code(void *code) {
	free(code);
	return PZ_OK;
}

int pz_module_load(const void *code, pz_module_t *module) {
	PZ_REQUIRE(code && module, "pz_module_load: null argument");
	hipModule_t mod;
	PZ_HIP(hipModuleLoadData(&mod, code));
	*module = new pz_module{mod};
	return PZ_OK;
}

int pz_module_unload(pz_module_t module) {
	if (!module) return PZ_OK;
	hipError_t e = hipModuleUnload(module->mod);
	delete module;
	if (e != hipSuccess) {
		pz::set_error("hipModuleUnload failed: %s", hipGetErrorString(e));
		return PZ_ERR_HIP;
	}
	return PZ_OK;
}

int pz_module_function(pz_module_t module, const char *name, void **function) {
	PZ_REQUIRE(module && name && function, "pz_module_function: null argument");
	hipFunction_t fn;
	hipError_t e = hipModuleGetFunction(&fn, module->mod, name);
	if (e != hipSuccess) {
		pz::set_error("the module has no kernel %s (%s)", name, hipGetErrorString(e));
		return PZ_ERR_INVALID;
	}
	*function = (void *)fn;
	return PZ_OK;
}

// Launches `function` with the kernel arguments packed in `args` the way the kernel's parameter list lays them out (each at its
// natural alignment; puzzlelib_amd/rtc.py packs them) — Function.__call__(*args, block=, grid=) of Cuda/Source/Core/Module.c:258-290.
int pz_function_launch(void *function, const unsigned *grid, const unsigned *block, unsigned shared_bytes, const void *args,
                       size_t args_bytes, pz_stream_t stream) {
	PZ_REQUIRE(function && grid && block && (args || args_bytes == 0), "pz_function_launch: null argument");
	PZ_REQUIRE(block[0] * block[1] * block[2] >= 1 && block[0] * block[1] * block[2] <= 1024, "pz_function_launch: block of %u x %u x %u threads",
	           block[0], block[1], block[2]);
	if (grid[0] == 0 || grid[1] == 0 || grid[2] == 0) return PZ_OK;
	size_t size = args_bytes;
	void *config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, const_cast<void *>(args), HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
	PZ_HIP(hipModuleLaunchKernel((hipFunction_t)function, grid[0], grid[1], grid[2], block[0], block[1], block[2], shared_bytes,
	                             pz::as_stream(stream), nullptr, config));
	return PZ_OK;
}

}  // extern "C"

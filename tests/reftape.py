"""
Record / replay of a test program at the backend object's surface.

The reference's own unit tests (Modules/Conv2D.py:80-353, Optimizers/Optimizer.py:249-325, Handlers/Trainer.py:38-108, ...) cannot
travel to the GPU box, and in the build container there is no GPU. So each one is run ONCE in the build container on top of this
repository's backend object with the C ABI emulated on host buffers (oracle/emu_cabi.py), where the reference's own asserts
judge the values — and everything the test did to the backend object is written down as a TAPE: every attribute it fetched,
every call it made (arguments by reference to earlier results, host arrays by value), every host value it read back, and the
point at which it dropped each object. `replay` runs a tape against a live backend object — on the MI355X: the real library
under the same Python glue, fusion policy included — and requires every value read back to equal the value the reference's
asserts accepted, within the fp32 tolerance stated in the tape.

A tape is data: a JSON list of operations plus the arrays it mentions (one .npz). It contains no reference source text.
Written by oracle/make_reftests.py (recorder), read by tests/test_gpu_7_reftests.py (replayer).

Large host inputs are not stored: the reference's tests draw them from numpy's global generator, so the recorder logs every
call of that generator (name, arguments, in order, from the seed) and an uploaded array that equals `call k's output,
converted to this dtype, rows lo:hi` is written down as exactly that; the replayer re-issues the calls on a private
RandomState (the legacy generator's streams are fixed across numpy versions) and rebuilds the array. That is what lets the
tests that push 40-500 MB of random data through a network (Handlers/Trainer.py:38-108, Containers/Sequential.py:243-298,
Models/Nets/*.py) travel as tapes of a few hundred KB. Tests that assert nothing about values (those same ones) additionally
get AUDITS: when the test drops a device array that is plainly stored (no pending description), a strided sample of ~1000 of
its elements is recorded, and the replayer requires the device's array to show the same sample at that point.

Values produced by the device's random generator differ between the emulation (numpy) and the device (Philox): after every
call of a generator method the recorder notes the values the filled array got ("poke") and the replayer writes them over
the device's own — the test then continues on identical numbers.
"""
import enum, json, weakref

import numpy as np

PLAIN = (bool, int, float, str, bytes, type(None), np.generic, np.dtype, slice, type(Ellipsis))
RNG_METHODS = ("fillUniform", "fillNormal", "fillInteger")


def isPlain(x):
	if isinstance(x, enum.Enum):
		return False
	if isinstance(x, PLAIN) or isinstance(x, np.ndarray):
		return True
	if isinstance(x, type) and issubclass(x, np.generic):
		return True
	if isinstance(x, (tuple, list)):
		return all(isPlain(v) for v in x)
	return False


# ---------------------------------------------------------------------------------------------------- recorder
RNG_MIN_BYTES = 1 << 10           # host arrays from this size on are looked up among the generators' outputs
AUDIT_SAMPLES = 256
AUDIT_DENSE, AUDIT_EVERY = 160, 40      # the first eligible drops are all audited, later ones every so often (a training loop drops thousands)

# What the EMULATED device generator (oracle/emu_cabi.py: pz_rng_fill_*) draws per fill, as functions of a numpy RandomState seeded
# with pz_rng_create's seed: written down here because recorder (the emulation) and replayer (which pokes these values over the
# device's own Philox output) must agree on them.
DEVICE_DRAWS = {
	"u32": lambda state, count: state.randint(0, 2 ** 32, size=int(count), dtype=np.uint64).astype(np.uint32),
	"uniform": lambda state, count: (1.0 - state.random_sample(int(count))).astype(np.float32),
	"normal": lambda state, count, mean, stddev: state.normal(mean, stddev, size=int(count)).astype(np.float32),
}


class LoggedState:
	"""the generator behind one emulated pz_rng handle: draws through DEVICE_DRAWS; a tape may listen (Tape.deviceState)"""

	def __init__(self, seed, listener=None):
		self.state, self.listener = np.random.RandomState(int(seed) & 0xffffffff), listener

	def draw(self, kind, *args):
		result = DEVICE_DRAWS[kind](self.state, *args)
		if self.listener is not None:
			self.listener(kind, args, result)
		return result


def auditSample(ary):
	"""a strided sample of a plainly stored, contiguous device array (None: not eligible — nothing may be forced by looking)"""
	try:
		from puzzlelib_amd import lazy
		if not hasattr(ary, "gpudata") or not ary.contiguous or ary.size == 0 or ary.dtype not in (np.float32, np.int32):
			return None
		if lazy.pending(ary) is not None or not lazy.quiet(ary):
			return None
		flat = ary.ravel()
		step = max(1, ary.size // AUDIT_SAMPLES)
		return flat[::step].get() if step > 1 else flat.get()
	except Exception:
		return None


class Tape:
	def __init__(self, name, atol=1e-5, rtol=1e-4, audit=False):
		self.name, self.ops, self.arrays = name, [], {}
		self.atol, self.rtol = atol, rtol
		self.nextId = 0
		self.live = weakref.WeakValueDictionary()       # id(real object) -> proxy
		self.closed = False
		self.audit, self.audits, self.eligible = audit, 0, 0
		# generator streams: "g" = numpy's global RandomState (its seed() calls are calls like any other); "d<i>" = the i-th
		# emulated device generator, {"seed": s, "calls": [...]}
		self.streams = {"g": {"seed": None, "calls": []}}
		self.rngCalls = self.streams["g"]["calls"]
		self.rngOut, self.rngCast, self.rngBroken = {}, {}, False

	def deviceState(self, seed):
		"""factory for oracle/emu_cabi.py's pz_rng_create while this tape records"""
		sid = "d%d" % (len(self.streams) - 1)
		calls = []
		self.streams[sid] = {"seed": int(seed) & 0xffffffff, "calls": calls}

		def listener(kind, args, result):
			if not self.closed:
				calls.append({"n": kind, "a": self.encPlain(list(args)), "kw": {"dict": {}}, "uses": 0})
				if result.nbytes >= RNG_MIN_BYTES:
					self.rngOut[(sid, len(calls) - 1)] = result
		return LoggedState(seed, listener)

	# ---- numpy's global generator (see the module comment)
	def watchGenerator(self):
		"""wraps every public function of numpy.random that draws from the global RandomState; returns the undo"""
		state = np.random.mtrand._rand
		saved = {}
		for fname in dir(state):
			bound = getattr(state, fname, None)
			if fname.startswith("_") or not callable(bound) or getattr(np.random, fname, None) is None or fname in ("get_state", "set_state"):
				continue
			saved[fname] = getattr(np.random, fname)
			setattr(np.random, fname, self.loggedGenerator(fname, saved[fname]))

		def undo():
			for fname, fn in saved.items():
				setattr(np.random, fname, fn)
		return undo

	def loggedGenerator(self, fname, fn):
		def call(*args, **kwargs):
			result = fn(*args, **kwargs)
			if self.closed or self.rngBroken:
				return result
			try:
				entry = {"n": fname, "a": self.encPlain(list(args)), "kw": self.encPlain(dict(kwargs)), "uses": 0}
			except TypeError:
				self.rngBroken = True        # (an array argument — shuffle in place ...): later inputs are stored by value
				return result
			self.rngCalls.append(entry)
			if isinstance(result, np.ndarray) and result.nbytes >= RNG_MIN_BYTES:
				self.rngOut[("g", len(self.rngCalls) - 1)] = result
			return result
		call.__name__ = fname
		return call

	def encPlain(self, x):
		if isinstance(x, (bool, int, float, str, type(None))):
			return x
		if isinstance(x, np.generic):
			return x.item()
		if isinstance(x, (tuple, list)):
			return {"tuple" if isinstance(x, tuple) else "list": [self.encPlain(v) for v in x]}
		if isinstance(x, dict):
			return {"dict": {str(k): self.encPlain(v) for k, v in x.items()}}
		if isinstance(x, np.dtype):
			return {"dtype": str(x)}
		if isinstance(x, type) and issubclass(x, np.generic):
			return {"nptype": np.dtype(x).name}
		raise TypeError("not a plain generator argument: %r" % (x, ))

	def matchGenerator(self, x):
		"""{"rng": [stream, k], "dt": dtype, "lo": a, "hi": b[, "shape": s]} when x == output k of a logged generator stream, cast
		to x's dtype, rows a:b (a flat output may be cut to x's shape)"""
		if x.ndim == 0 or x.size == 0:
			return None
		for key in sorted(self.rngOut, key=lambda sk: (sk[0] != "g", -sk[1])):
			out = self.rngOut[key]
			shape = None
			y = x
			if out.ndim == 1 and x.ndim > 1:
				shape, y = list(x.shape), x.reshape(-1)
			if out.ndim != y.ndim or out.shape[1:] != y.shape[1:] or out.shape[0] < y.shape[0]:
				continue
			cast = self.rngCast.get((key, y.dtype.str))
			if cast is None:
				with np.errstate(all="ignore"):
					cast = self.rngCast[(key, y.dtype.str)] = out.astype(y.dtype) if out.dtype != y.dtype else out
			flat, rows = cast.reshape(cast.shape[0], -1), y.reshape(y.shape[0], -1)
			for lo in np.flatnonzero(flat[:, 0] == rows[0, 0])[:64]:
				lo = int(lo)
				if lo + y.shape[0] <= cast.shape[0] and np.array_equal(cast[lo:lo + y.shape[0]], y):
					self.streams[key[0]]["calls"][key[1]]["uses"] += 1
					enc = {"rng": [key[0], key[1]], "dt": y.dtype.str, "lo": lo, "hi": lo + y.shape[0]}
					if shape is not None:
						enc["shape"] = shape
					return enc
		return None

	def encArray(self, x):
		"""a host array on the tape: a constant, a piece of a generator's output, or its values"""
		if x.nbytes >= RNG_MIN_BYTES and x.size > 0:
			first = x.reshape(-1)[0]
			if (x == first).all() and (first == first):
				return {"const": first.item(), "dt": x.dtype.str, "shape": list(x.shape)}
			if self.rngOut:
				drawn = self.matchGenerator(np.ascontiguousarray(x))
				if drawn is not None:
					return drawn
		return {"np": self.store(x)}

	def newId(self):
		self.nextId += 1
		return self.nextId

	def store(self, array):
		key = "a%d" % len(self.arrays)
		self.arrays[key] = np.array(array, copy=True)
		return key

	def emit(self, **op):
		if not self.closed:
			self.ops.append(op)

	# ---- encoding of arguments / results
	def enc(self, x):
		if isinstance(x, Proxy):
			return {"ref": object.__getattribute__(x, "_pid")}
		if isinstance(x, np.ndarray):
			return self.encArray(x)
		if isinstance(x, np.generic):
			return {"s": x.item(), "dt": str(x.dtype)}
		if isinstance(x, np.dtype):
			return {"dtype": str(x)}
		if isinstance(x, type) and issubclass(x, np.generic):
			return {"nptype": np.dtype(x).name}
		if isinstance(x, slice):
			return {"slice": [self.enc(x.start), self.enc(x.stop), self.enc(x.step)]}
		if x is Ellipsis:
			return {"ellipsis": 1}
		if isinstance(x, tuple):
			return {"tuple": [self.enc(v) for v in x]}
		if isinstance(x, list):
			return {"list": [self.enc(v) for v in x]}
		if isinstance(x, dict):
			return {"dict": {str(k): self.enc(v) for k, v in x.items()}}
		if isinstance(x, (bool, int, float, str, type(None))):
			return x
		if hasattr(x, "typegen"):          # a C type object of the reference's code generator (Compiler/Codegen/Types.py): its spelling
			return {"ctype": str(x)}
		raise TypeError("cannot put %r (%s) on a tape" % (x, type(x)))

	def wrap(self, real):
		"""result of an attribute fetch / call as the test sees it, and its encoding"""
		if isPlain(real):
			if isinstance(real, np.ndarray):
				return real, {"np": self.store(real)}
			if isinstance(real, (float, np.floating)):
				return real, {"f": float(real)}
			return real, {"plain": 1}
		if isinstance(real, (tuple, list)):
			pairs = [self.wrap(v) for v in real]
			return type(real)(p[0] for p in pairs), {"tuple" if isinstance(real, tuple) else "list": [p[1] for p in pairs]}
		if isinstance(real, dict):
			pairs = {k: self.wrap(v) for k, v in real.items()}
			return {k: p[0] for k, p in pairs.items()}, {"dict": {str(k): p[1] for k, p in pairs.items()}}
		proxy = self.live.get(id(real))
		if proxy is None or object.__getattribute__(proxy, "_real") is not real:
			proxy = Proxy(self, real, self.newId())
			try:
				self.live[id(real)] = proxy
			except TypeError:
				pass
		return proxy, {"ref": object.__getattribute__(proxy, "_pid")}

	def scalarTolerance(self):
		return scalarTolerance(self.ops, self.atol, self.rtol)

	def save(self, path):
		self.closed = True
		header = {"name": self.name, "atol": self.atol, "rtol": self.rtol, "scalars": self.scalarTolerance(), "ops": self.ops}
		streams = {}
		for sid, stream in self.streams.items():
			used = [k for k, call in enumerate(stream["calls"]) if call["uses"]]
			if used:
				streams[sid] = {"seed": stream["seed"], "calls": stream["calls"][:max(used) + 1]}
		if streams:
			header["rng"] = streams
		np.savez_compressed(path, __tape__=np.frombuffer(json.dumps(header, separators=(",", ":")).encode(), dtype=np.uint8), **self.arrays)


EARLY = 20


def scalarTolerance(ops, atol, rtol):
	"""Host scalars a test reads back (Blas.dot / cost errors: "f" results). A test that reads many of them follows a training
	TRAJECTORY — Optimizers/Optimizer.py:249-325 trainSimpleTest / trainHardTest print the error of 200 consecutive updates and
	assert nothing about it — and a trajectory amplifies rounding differences step by step: RMSProp's error starts at 5e-3 and
	oscillates between 6e-5 and 1.2e-4 after 150 updates, where two fp32 summation orders differ by 12 %. The first EARLY such
	monitors of a trajectory tape are held to atol + 1e-3 |x| (nothing has been amplified yet: a wrong kernel shows there), the
	later ones to 2 % of the LARGEST value the tape reads. The arrays a test reads — what the reference's asserts compare — and
	the scalars of tapes without a trajectory keep atol / rtol."""
	scalars = []

	def walk(enc):
		if isinstance(enc, dict):
			if "f" in enc:
				scalars.append(abs(enc["f"]))
			for key in ("tuple", "list"):
				for item in enc.get(key, ()):
					walk(item)
	for op in ops:
		if op["k"] in ("attr", "call"):
			walk(op["r"])
	if len(scalars) <= EARLY:
		return {"atol": atol, "rtol": rtol, "trajectory": False}
	return {"atol": atol, "rtol": 1e-3, "trajectory": True, "late_atol": max(atol, 2e-2 * max(scalars))}


def unwrap(x):
	if isinstance(x, Proxy):
		return object.__getattribute__(x, "_real")
	if isinstance(x, tuple):
		return tuple(unwrap(v) for v in x)
	if isinstance(x, list):
		return [unwrap(v) for v in x]
	if isinstance(x, dict):
		return {k: unwrap(v) for k, v in x.items()}
	return x


class Proxy:
	"""stands for one object of the backend (the backend itself, a class, a context, an array, a kernel, a bound method ...)"""
	__slots__ = ("_tape", "_real", "_pid", "_name", "__weakref__")

	def __init__(self, tape, real, pid, name=None):
		object.__setattr__(self, "_tape", tape)
		object.__setattr__(self, "_real", real)
		object.__setattr__(self, "_pid", pid)
		object.__setattr__(self, "_name", name)

	def __del__(self):
		try:
			tape = object.__getattribute__(self, "_tape")
			if tape.audit and not tape.closed and hasattr(object.__getattribute__(self, "_real"), "gpudata"):
				tape.eligible += 1
				if tape.eligible <= AUDIT_DENSE or tape.eligible % AUDIT_EVERY == 0:
					sample = auditSample(object.__getattribute__(self, "_real"))
					if sample is not None and np.isfinite(sample).all():
						tape.audits += 1
						tape.emit(k="audit", id=object.__getattribute__(self, "_pid"), v=tape.store(sample))
			tape.emit(k="del", id=object.__getattribute__(self, "_pid"))
		except Exception:
			pass

	def __getattr__(self, name):
		tape, real, pid = (object.__getattribute__(self, n) for n in ("_tape", "_real", "_pid"))
		value = getattr(real, name)
		if isPlain(value):
			return value
		seen, encoded = tape.wrap(value)
		tape.emit(k="attr", t=pid, n=name, r=encoded)
		if isinstance(seen, Proxy):
			object.__setattr__(seen, "_name", name)
		return seen

	def __setattr__(self, name, value):
		tape, real, pid = (object.__getattribute__(self, n) for n in ("_tape", "_real", "_pid"))
		tape.emit(k="setattr", t=pid, n=name, v=tape.enc(value))
		setattr(real, name, unwrap(value))

	def __call__(self, *args, **kwargs):
		tape, real, pid, name = (object.__getattribute__(self, n) for n in ("_tape", "_real", "_pid", "_name"))
		a, kw = tape.enc(list(args)), tape.enc(dict(kwargs))
		result = real(*unwrap(args), **unwrap(kwargs))
		seen, encoded = tape.wrap(result)
		tape.emit(k="call", t=pid, a=a, kw=kw, r=encoded)
		if name in RNG_METHODS:
			for arg in list(args) + list(kwargs.values()):
				if isinstance(arg, Proxy) and hasattr(unwrap(arg), "get") and hasattr(unwrap(arg), "set"):
					tape.emit(k="poke", t=object.__getattribute__(arg, "_pid"), v=tape.encArray(unwrap(arg).get()))
		return seen

	def __instancecheck__(self, obj):
		return isinstance(unwrap(obj), object.__getattribute__(self, "_real"))

	def __iter__(self):
		tape, real = object.__getattribute__(self, "_tape"), object.__getattribute__(self, "_real")
		if isinstance(real, type) and issubclass(real, enum.Enum):
			return iter([getattr(self, member.name) for member in real])
		raise TypeError("iteration over %r is not taped" % (real, ))

	def __repr__(self):
		return "<taped %r>" % (object.__getattribute__(self, "_real"), )

	def __bool__(self):
		return True


def _dunder(name):
	def method(self, *args):
		return Proxy.__getattr__(self, name)(*args)
	method.__name__ = name
	return method


for _n in ("__getitem__", "__setitem__", "__add__", "__radd__", "__iadd__", "__mul__", "__rmul__", "__imul__", "__sub__", "__len__",
		   "__enter__", "__exit__", "__eq__", "__hash__"):
	if _n in ("__eq__", "__hash__"):
		continue
	setattr(Proxy, _n, _dunder(_n))


# ---------------------------------------------------------------------------------------------------- replayer
def load(path):
	data = np.load(path, allow_pickle=False)
	header = json.loads(bytes(data["__tape__"]).decode())
	return header, data


def replay(path, getBackend, check=None):
	"""runs the tape at `path` against the backend `getBackend(initmode)` returns; `check(got, want, what)` compares a value read
	back with the recorded one (default: |got - want| <= atol + rtol |want| with the tape's tolerances). Returns the number of
	values compared."""
	header, data = load(path)
	atol, rtol = header["atol"], header["rtol"]
	refs, compared = {}, [0]

	# the generator calls of the recording, re-issued on demand and in order; an output lives until its last use
	streams = header.get("rng", {})
	drawn, left, issued, states = {}, {}, {}, {}
	for sid, stream in streams.items():
		issued[sid] = 0
		states[sid] = np.random.RandomState() if stream["seed"] is None else np.random.RandomState(stream["seed"])
		for k, call in enumerate(stream["calls"]):
			left[(sid, k)] = call["uses"]

	def plain(x):
		if isinstance(x, dict):
			if "tuple" in x:
				return tuple(plain(v) for v in x["tuple"])
			if "list" in x:
				return [plain(v) for v in x["list"]]
			if "dict" in x:
				return {k: plain(v) for k, v in x["dict"].items()}
			if "dtype" in x:
				return np.dtype(x["dtype"])
			if "nptype" in x:
				return np.dtype(x["nptype"]).type
		return x

	def generated(sid, k):
		calls, state = streams[sid]["calls"], states[sid]
		while issued[sid] <= k:
			call = calls[issued[sid]]
			if sid == "g":
				result = getattr(state, call["n"])(*plain(call["a"]), **plain(call["kw"]))
			else:
				result = DEVICE_DRAWS[call["n"]](state, *plain(call["a"]))
			if call["uses"]:
				drawn[(sid, issued[sid])] = result
			issued[sid] += 1
		result = drawn[(sid, k)]
		left[(sid, k)] -= 1
		if left[(sid, k)] == 0:
			del drawn[(sid, k)]
		return result

	def array(x):
		"""a host array of the tape (Tape.encArray)"""
		if "np" in x:
			return data[x["np"]]
		if "const" in x:
			return np.full(x["shape"], x["const"], dtype=np.dtype(x["dt"]))
		with np.errstate(all="ignore"):
			piece = np.ascontiguousarray(generated(*x["rng"]).astype(np.dtype(x["dt"]), copy=False)[x["lo"]:x["hi"]])
		return piece.reshape(x["shape"]) if "shape" in x else piece

	scalars = scalarTolerance(header["ops"], atol, rtol)
	nscalar = [0]

	def default_check(got, want, what, scalar=False):
		got, want = np.asarray(got), np.asarray(want)
		assert got.shape == want.shape, "%s: shape %s, recorded %s" % (what, got.shape, want.shape)
		if want.dtype.kind == "f":
			# infinities and NaNs (log-space forward variables of impossible CTC paths ...) must sit where they were recorded
			odd = ~np.isfinite(want)
			if odd.any():
				assert np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(np.asarray(got)[odd & ~np.isnan(want)], want[odd & ~np.isnan(want)]), \
					"%s: non-finite values differ from the recorded ones" % what
				got, want = np.where(odd, 0.0, got), np.where(odd, 0.0, want)
			err = np.abs(got.astype(np.float64) - want.astype(np.float64))
			a, r = atol, rtol
			if scalar:
				nscalar[0] += 1
				late = scalars["trajectory"] and nscalar[0] > EARLY
				a, r = (scalars["late_atol"] if late else scalars["atol"]), scalars["rtol"]
			bound = a + r * np.abs(want.astype(np.float64))
			bad = ~(err <= bound)
			assert not bad.any(), "%s: %d of %d values off, worst |err| %.3e where the recorded value is %.3e (bound %.1e + %.1e |x|)" % (
				what, int(bad.sum()), bad.size, float(err[bad].max()), float(np.abs(want)[bad][np.argmax(err[bad])]),
				a, r)
		else:
			assert np.array_equal(got, want), "%s: integer values differ" % what
	if check is None:
		check = default_check
	else:
		user = check
		check = lambda got, want, what, scalar=False: user(got, want, what)

	def dec(x):
		if isinstance(x, dict):
			if "ref" in x:
				return refs[x["ref"]]
			if "np" in x or "rng" in x or "const" in x:
				return array(x)
			if "s" in x:
				return np.dtype(x["dt"]).type(x["s"])
			if "dtype" in x:
				return np.dtype(x["dtype"])
			if "nptype" in x:
				return np.dtype(x["nptype"]).type
			if "ctype" in x:
				return x["ctype"]
			if "slice" in x:
				return slice(*[dec(v) for v in x["slice"]])
			if "ellipsis" in x:
				return Ellipsis
			if "tuple" in x:
				return tuple(dec(v) for v in x["tuple"])
			if "list" in x:
				return [dec(v) for v in x["list"]]
			if "dict" in x:
				return {k: dec(v) for k, v in x["dict"].items()}
			raise ValueError("unknown tape value %r" % (x, ))
		return x

	def bind(enc, value, what):
		"""binds the references of a recorded result to the live result; compares recorded host values"""
		if not isinstance(enc, dict):
			return
		if "ref" in enc:
			refs[enc["ref"]] = value
		elif "np" in enc:
			compared[0] += 1
			check(value, data[enc["np"]], what)
		elif "f" in enc:
			compared[0] += 1
			check(np.float64(value), np.float64(enc["f"]), what, True)
		elif "tuple" in enc or "list" in enc:
			items = enc.get("tuple", enc.get("list"))
			assert len(value) == len(items), "%s: %d results, recorded %d" % (what, len(value), len(items))
			for i, (e, v) in enumerate(zip(items, value)):
				bind(e, v, "%s[%d]" % (what, i))
		elif "dict" in enc:
			for k, e in enc["dict"].items():
				bind(e, value[k], "%s[%s]" % (what, k))

	names, audited = {}, [0]
	for i, op in enumerate(header["ops"]):
		kind = op["k"]
		if kind == "root":
			refs[op["r"]] = getBackend(op["initmode"])
		elif kind == "attr":
			names[op["r"].get("ref", -1) if isinstance(op["r"], dict) else -1] = op["n"]
			bind(op["r"], getattr(refs[op["t"]], op["n"]), "op %d: .%s" % (i, op["n"]))
		elif kind == "setattr":
			setattr(refs[op["t"]], op["n"], dec(op["v"]))
		elif kind == "call":
			what = "op %d: %s(...)" % (i, names.get(op["t"], "?"))
			bind(op["r"], refs[op["t"]](*dec(op["a"]), **dec(op["kw"])), what)
		elif kind == "poke":
			refs[op["t"]].set(array(op["v"]) if isinstance(op["v"], dict) else data[op["v"]])
		elif kind == "audit":
			# a sample of an array the test is about to drop (tests that assert nothing about values): same elements, same values —
			# to 1e-3 of the sample's largest magnitude (these arrays sit behind whole networks of unnormalised layers)
			live = refs.get(op["id"])
			sample = auditSample(live) if live is not None else None
			if sample is not None:
				want = data[op["v"]]
				assert sample.shape == want.shape, "op %d: audit of a %s array, recorded %s" % (i, sample.shape, want.shape)
				if want.dtype.kind == "f":
					# (the dense early audits sit in front of any amplification; the sparse late ones of a training loop follow a
					# trajectory hundreds of updates long — held like the late scalar monitors, to 2 % of the sample's top)
					audited[0] += 1
					rel = 1e-3 if audited[0] <= AUDIT_DENSE else 2e-2
					scale = float(np.abs(want).max())
					err = float(np.abs(sample.astype(np.float64) - want.astype(np.float64)).max()) if want.size else 0.0
					assert err <= 1e-5 + rel * scale, "op %d: audit of %s differs by %.3e (largest recorded magnitude %.3e)" % (i, names.get(op["id"], "?"), err, scale)
				else:
					assert np.array_equal(sample, want), "op %d: audit of an integer array differs" % i
				compared[0] += 1
		elif kind == "del":
			refs.pop(op["id"], None)
			names.pop(op["id"], None)
		else:
			raise ValueError("unknown tape operation %r" % kind)
	return compared[0]

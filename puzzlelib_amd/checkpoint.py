"""
Checkpoints of the stand-alone harness: a network's parameters and attributes (and, optionally, an optimizer's state) to
a file and back.

The reference stores a module tree in HDF5 (Modules/Module.py:179-283): group `params` holds the tensors under running
indices, group `links` maps "<module path>.<param>" to an index (so tied variables are stored once), group `attrs` holds
"<module path>.<attr>" (batch-norm running statistics). The same three-part structure is kept here — `params/<idx>`,
`links/<layer>.<param>` -> idx, `attrs/<layer>.<attr>`, plus `optimizer/...` and `meta/...` — in a numpy .npz container
(h5py is not part of this image); with h5py importable, `format="hdf5"` writes the identical structure as HDF5 groups and
datasets. `layer` is the spec's layer name (= the reference's module name for the shipped networks).

`load` also reads what the REFERENCE writes: "<net>.<module>.<field>" entry names (assumeUniqueNames=True, the form
Models/Nets/ResNet.py:118-119 loads) or container-qualified ones, no `meta` entry — tests/golden/refckpt_mini_*.npz are
the reference's own Module.save output (oracle/make_checkpoint_fixture.py), tests/test_gpu_5_nets.py loads them.

What a round trip must restore for training to continue bit for bit: parameters, running mean / variance, the batch-norm
layers' pass counters (their momentum factor is 1/passes until it reaches minFactor, Modules/BatchNormND.py:55-58), the
optimizer's step count and state tensors.
"""
import json

import numpy as np


def collect(net, optimizer=None):
	"""-> {path: ndarray} with the structure described above"""
	out, index = {}, {}
	for name, param in net.namedParams().items():
		key = id(param)
		if key not in index:
			index[key] = len(index)
			out["params/%d" % index[key]] = param.data.get()
		out["links/" + name] = np.array(index[key], dtype=np.int64)
	for name, attr in net.namedAttrs().items():
		out["attrs/" + name] = attr.get()

	meta = {"name": net.name, "bn_passes": {l.name: l.cfg["passes"] for l in net.walk() if l.kind == "bn"}}
	if optimizer is not None:
		meta["optimizer"] = {"rule": optimizer.rule, "t": optimizer.t, "flat": optimizer.params is not None}
		if optimizer.params is not None:
			meta["optimizer"]["order"] = list(optimizer.params.blocks.keys())
		for idx, (target, state) in enumerate(optimizer.targets):
			for key, tensor in state.items():
				out["optimizer/%d/%s" % (idx, key)] = tensor.get()
	out["meta/json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
	return out


def save(net, path, optimizer=None, format="npz"):
	tensors = collect(net, optimizer)
	if format == "npz":
		with open(path, "wb") as f:           # a file object: np.savez would append ".npz" to a bare path and load() miss it
			np.savez(f, **{k.replace("/", "|"): v for k, v in tensors.items()})
	elif format == "hdf5":
		import h5py
		with h5py.File(path, "w") as hdf:
			for key, value in tensors.items():
				hdf.create_dataset(key, data=value)
	else:
		raise ValueError(format)


def read(path):
	with open(path, "rb") as f:               # the container is recognised by its magic, not by the file's name
		magic = f.read(8)
	if magic == b"\x89HDF\r\n\x1a\n":
		import h5py
		out = {}
		with h5py.File(path, "r") as hdf:
			hdf.visititems(lambda name, obj: out.__setitem__(name, np.array(obj)) if hasattr(obj, "shape") else None)
		return out
	with np.load(path) as z:
		return {k.replace("|", "/"): z[k] for k in z.files}


def resolver(tensors, group, path):
	"""-> function("<layer>.<field>") -> key of that entry in `group`.

	This package writes "<layer>.<field>". The reference writes "<net>.<module>.<field>" with assumeUniqueNames=True (how
	Models/Nets/ResNet.py:118-119 loads) and "<net>.<container>. ... .<module>.<field>" without (Modules/Module.py:187-203,
	Containers/Container.py:149-153): its entries are found by their last two name components, which is exactly the
	identity assumeUniqueNames relies on; an entry name claimed by two different modules is refused."""
	prefix = group + "/"
	exact, tails = {}, {}
	for key in tensors:
		if key.startswith(prefix):
			name = key[len(prefix):]
			exact[name] = key
			tails.setdefault(".".join(name.split(".")[-2:]), []).append(key)

	def find(name):
		if name in exact:
			return exact[name]
		hits = tails.get(name, [])
		if len(hits) == 1:
			return hits[0]
		if not hits:
			raise KeyError("checkpoint %s has no entry %s%s" % (path, prefix, name))
		raise KeyError("checkpoint %s: %s%s is ambiguous (%s)" % (path, prefix, name, ", ".join(sorted(hits))))
	return find


def load(net, path, optimizer=None):
	"""Restores `net` (and `optimizer`, which must already be set up on it the same way) from a checkpoint — one of this
	package's, or a file the reference's Module.save / Container.save wrote (Modules/Module.py:179-283: groups params / links /
	attrs, no meta entry; batch-norm pass counters then start at 0 as they do in a freshly loaded reference module, whose
	numOfProps is not saved, Modules/BatchNormND.py:33-58)."""
	tensors = read(path)
	meta = json.loads(bytes(tensors["meta/json"]).decode()) if "meta/json" in tensors else {}

	link, attrOf = resolver(tensors, "links", path), resolver(tensors, "attrs", path)
	for name, param in net.namedParams().items():
		value = tensors["params/%d" % int(tensors[link(name)])]
		if value.shape != param.data.shape:
			raise ValueError("parameter %s: checkpoint shape %s, network shape %s" % (name, value.shape, param.data.shape))
		param.data.set(np.ascontiguousarray(value.astype(np.float32, casting="safe", copy=False)))
	for name, attr in net.namedAttrs().items():
		value = tensors[attrOf(name)]
		if value.shape != attr.shape:
			raise ValueError("attribute %s: checkpoint shape %s, network shape %s" % (name, value.shape, attr.shape))
		attr.set(np.ascontiguousarray(value.astype(np.float32, casting="safe", copy=False)))
	passes = meta.get("bn_passes", {})
	for layer in net.walk():
		if layer.kind == "bn":
			layer.cfg["passes"] = int(passes.get(layer.name, 0))

	if optimizer is not None:
		saved = meta.get("optimizer")
		if saved is None or saved["rule"] != optimizer.rule or saved["flat"] != (optimizer.params is not None):
			raise ValueError("checkpoint %s holds no matching optimizer state" % path)
		if saved["flat"] and saved["order"] != list(optimizer.params.blocks.keys()):
			raise ValueError("the flat arena is laid out differently from the checkpoint's")
		optimizer.t = int(saved["t"])
		for idx, (target, state) in enumerate(optimizer.targets):
			for key, tensor in state.items():
				tensor.set(tensors["optimizer/%d/%s" % (idx, key)])
	return meta

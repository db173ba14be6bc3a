"""
TEST INFRASTRUCTURE (build container only) — writes what the REFERENCE writes when it saves a network, so that
puzzlelib_amd/checkpoint.py can be held to reading it.

The imported reference (oracle/refimport.py, numpy CPU backend) builds a small ResNet out of its OWN builders
(Models/Nets/ResNet.py:23-66 residBlock / residMiniBlock; the same modules and naming as loadResNet, :69-121, at small
widths so that the fixture stays a few kilobytes), gives every parameter and every running statistic a distinct value and
calls its own Module.save (Containers/Container.py:138-170 -> Modules/Module.py:179-231) twice — with assumeUniqueNames=True,
as Models/Nets/ResNet.py:118-119 loads ("<net>.<module>.<param>"), and without ("<net>.<container path>.<module>.<param>").
h5py is not part of this image: save() is handed an in-memory stand-in for the open file (it accepts any object,
Module.py:370-395) that records exactly the calls the reference makes on it — require_group, create_dataset, item
assignment — and the recorded tree is written as the .npz mirror of the HDF5 layout ("group|dataset" keys, the container of
checkpoint.save(format="npz")). The reference's own evaluation-mode forward pass on a fixed input is stored next to it.

    python oracle/make_checkpoint_fixture.py            # rewrites tests/golden/refckpt_*.npz
    python oracle/make_checkpoint_fixture.py --check    # the committed files still are what the reference writes
"""
import os, sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")


class Group(dict):
	"""the few h5py.Group calls Module.save / Container.save make"""
	def require_group(self, name):
		return self.setdefault(name, Group())

	def create_dataset(self, name, shape=None, dtype=None, data=None, compression=None):
		self[name] = np.array(data)


class TreeFile(Group):
	def flush(self):
		pass

	def close(self):
		pass


def flatten(tree, prefix=""):
	out = {}
	for key, value in tree.items():
		if isinstance(value, Group):
			out.update(flatten(value, prefix + key + "|"))
		else:
			out[prefix + key] = np.asarray(value)
	return out


def build():
	import refimport
	refimport.setup()
	from PuzzleLib.Backend import gpuarray
	from PuzzleLib.Containers.Sequential import Sequential
	from PuzzleLib.Modules.Conv2D import Conv2D
	from PuzzleLib.Modules.BatchNorm2D import BatchNorm2D
	from PuzzleLib.Modules.Activation import Activation, relu
	from PuzzleLib.Modules.MaxPool2D import MaxPool2D
	from PuzzleLib.Modules.AvgPool2D import AvgPool2D
	from PuzzleLib.Modules.Flatten import Flatten
	from PuzzleLib.Modules.Linear import Linear
	from PuzzleLib.Models.Nets.ResNet import residBlock

	np.random.seed(20260929)
	net = Sequential(name="ResNet-mini")
	net.append(Conv2D(3, 8, 7, stride=2, pad=3, name="conv1", initscheme="gaussian", useBias=False))
	net.append(BatchNorm2D(8, name="bn_conv1"))
	net.append(Activation(relu, name="conv1_relu"))
	net.append(MaxPool2D(3, 2, name="pool1"))
	net.extend(residBlock(8, 4, 1, "2a", True, False, False, "gaussian"))
	net.extend(residBlock(16, 4, 1, "2b", False, False, False, "gaussian"))
	net.extend(residBlock(16, 8, 2, "3a", True, False, False, "gaussian"))
	net.append(AvgPool2D(4, 1))
	net.append(Flatten())
	net.append(Linear(32, 10, initscheme="gaussian", name="fc10"))
	# (loadResNet ends in a SoftMax, which has no parameters and which the reference's numpy CPU backend cannot run)

	# distinct values everywhere, so that a tensor landing in the wrong place changes the output
	rng = np.random.RandomState(7)

	def visit(mod):
		for name, var in getattr(mod, "vars", {}).items():
			scale = 0.3 if name in ("W", "scale") else 0.1
			base = 1.0 if name == "scale" else 0.0
			var.data.set((base + scale * rng.randn(*var.data.shape)).astype(np.float32))
		for name, attr in getattr(mod, "attrs", {}).items():
			if hasattr(attr, "set"):
				val = 0.2 * rng.randn(*attr.shape) if name == "mean" else 0.5 + rng.rand(*attr.shape)
				attr.set(val.astype(np.float32))
		for sub in getattr(mod, "modules", {}).values():
			visit(sub)
	visit(net)

	out = {}
	for tag, unique in (("unique", True), ("full", False)):
		tree = TreeFile()
		net.save(hdf=tree, assumeUniqueNames=unique)
		out[tag] = flatten(tree)

	data = rng.randn(2, 3, 32, 32).astype(np.float32)
	net.evalMode()
	scores = net(gpuarray.to_gpu(data)).get()
	out["io"] = {"data": data, "scores": np.asarray(scores, dtype=np.float32)}
	return out


def resnet_entry_names():
	"""{"50" | "101" | "152": [[entry name, shape], ...]}: the link / attr entries the reference's loadResNet networks save with
	assumeUniqueNames=True (Models/Nets/ResNet.py:69-121), without the tensors."""
	import refimport
	refimport.setup()
	from PuzzleLib.Models.Nets.ResNet import loadResNet
	out = {}
	for layers in ("50", "101", "152"):
		net = loadResNet(None, layers, initscheme="none")
		names = []

		def visit(mod, path):
			for name, var in getattr(mod, "vars", {}).items():
				names.append(["links", "%s.%s.%s" % (net.name, mod.name, name), list(var.data.shape)])
			for name, attr in getattr(mod, "attrs", {}).items():
				if hasattr(attr, "shape"):
					names.append(["attrs", "%s.%s.%s" % (net.name, mod.name, name), list(attr.shape)])
			for sub in getattr(mod, "modules", {}).values():
				visit(sub, path + [sub.name])
		visit(net, [])
		out[layers] = names
	return out


def main():
	import json
	check = "--check" in sys.argv
	names, path = resnet_entry_names(), os.path.join(GOLDEN, "resnet_entry_names.json")
	if check:
		assert json.load(open(path)) == names, "%s differs from the reference's entry names" % path
		print("resnet_entry_names: unchanged (%s entries)" % ", ".join("%s: %d" % (k, len(v)) for k, v in names.items()))
	else:
		json.dump(names, open(path, "w"), separators=(",", ":"))
		print("wrote %s" % path)
	files = build()
	for tag, tensors in files.items():
		path = os.path.join(GOLDEN, "refckpt_mini_%s.npz" % tag)
		if check:
			with np.load(path) as z:
				assert sorted(z.files) == sorted(tensors), "%s: keys differ from what the reference writes now" % path
				for key in z.files:
					assert np.array_equal(z[key], tensors[key]), "%s: %s differs" % (path, key)
			print("refckpt_mini_%s: unchanged (%d entries)" % (tag, len(tensors)))
		else:
			with open(path, "wb") as f:
				np.savez(f, **tensors)
			print("wrote %s (%d entries)" % (path, len(tensors)))


if __name__ == "__main__":
	main()

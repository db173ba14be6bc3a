"""
Network definitions used by the reference's configs, as plain data ("specs") plus builders that turn a
spec into a module graph on the MI355X backend.

* LeNet       — Models/Nets/LeNet.py:13-33 (config 1, TestLib/CnnMnistLenet.py)
* NiN         — TestLib/CnnCifar10NIN.py:13-49 (config 3)
* ResNet-50   — Models/Nets/ResNet.py:23-121 (configs 4/5). Note the reference variant: MaxPool2D(3, 2) with
                pad 0 (55x55 stage-2 maps, ResNet.py:93) and the stride on the first 1x1 of a down-sampling
                block (ResNet.py:42).

Spec entries are documented in oracle/cpu_net.py (the CPU checker consumes the same data; it shares no
code with this package). Layer names are the reference's module names (the block ReLUs and the containers have no
counterpart object here, so parameter paths are "<layer>.<param>", not the reference's container-qualified ones).
"""
import string


def lenet_spec():
	# unnamed modules get their index in the container as name (Containers/Container.py:28-29)
	return [
		("conv", "0", 1, 16, 3, 1, 0, True),
		("maxpool", "1", 2, 2, 0),
		("relu", "2"),
		("conv", "3", 16, 32, 4, 1, 0, True),
		("maxpool", "4", 2, 2, 0),
		("relu", "5"),
		("flatten", "6"),
		("linear", "7", 32 * 5 * 5, 1024),
		("relu", "8"),
		("linear", "9", 1024, 10),
	]


def nin_spec():
	def cr(name, cin, cout, size, pad, relu):
		return [("conv", name, cin, cout, size, 1, pad, True), ("relu", relu)]

	spec = []
	spec += cr("conv1", 3, 192, 5, 2, "relu1")
	spec += cr("cccp1", 192, 160, 1, 0, "relu_cccp1")
	spec += cr("cccp2", 160, 96, 1, 0, "relu_cccp2")
	spec += [("maxpool", "pool1", 3, 2, 1), ("dropout", "drop3", 0.5)]
	spec += cr("conv2", 96, 192, 5, 2, "relu2")
	spec += cr("cccp3", 192, 192, 1, 0, "relu_cccp3")
	spec += cr("cccp4", 192, 192, 1, 0, "relu_cccp4")
	spec += [("avgpool", "pool2", 3, 2, 1), ("dropout", "drop6", 0.5)]
	spec += cr("conv3", 192, 192, 3, 1, "relu3")
	spec += cr("cccp5", 192, 192, 1, 0, "relu_cccp5")
	spec += cr("cccp6", 192, 10, 1, 0, "relu_cccp6")
	spec += [("avgpool", "pool3", 8, 1, 0), ("flatten", "flatten")]
	return spec


def _mini(inmaps, outmaps, size, stride, pad, blockname, mininame, act):
	layers = [
		("conv", "res%s_branch%s" % (blockname, mininame), inmaps, outmaps, size, stride, pad, False),
		("bn", "bn%s_branch%s" % (blockname, mininame), outmaps),
	]
	if act:
		layers.append(("relu", "res%s_branch%s_relu" % (blockname, mininame)))
	return layers


def _resid(inmaps, hmaps, stride, blockname, convShortcut):
	branch = _mini(inmaps, hmaps, 1, stride, 0, blockname, "2a", True) + \
		_mini(hmaps, hmaps, 3, 1, 1, blockname, "2b", True) + \
		_mini(hmaps, 4 * hmaps, 1, 1, 0, blockname, "2c", False)

	shortcut = _mini(inmaps, 4 * hmaps, 1, stride, 0, blockname, "1", False) if convShortcut else []
	return [("resid", branch, shortcut), ("relu", "res%s_relu" % blockname)]


def resnet_blocknames(layers):
	"""Block names per level as Models/Nets/ResNet.py:70-92 spells them: letters for ResNet-50 ("3a" ... "3d"), "<level>a" +
	"<level>b<number>" for the long stages of ResNet-101 / -152."""
	if layers == "50":
		level3, level4 = ["3%s" % a for a in string.ascii_lowercase[1:4]], ["4%s" % a for a in string.ascii_lowercase[1:6]]
	elif layers == "101":
		level3, level4 = ["3b%d" % i for i in range(1, 4)], ["4b%d" % i for i in range(1, 23)]
	elif layers == "152":
		level3, level4 = ["3b%d" % i for i in range(1, 8)], ["4b%d" % i for i in range(1, 36)]
	else:
		raise ValueError("Unsupported ResNet layers mode")
	return {2: ["2a", "2b", "2c"], 3: ["3a"] + level3, 4: ["4a"] + level4, 5: ["5a", "5b", "5c"]}


def resnet_spec(stages=((64, 3), (128, 4), (256, 6), (512, 3)), classes=1000, stem=64, softmax=True, blocknames=None):
	"""ResNet-50 by default; smaller `stages` give the mini-ResNets used by the parity tests; `blocknames` ({level: [name, ...]},
	resnet_blocknames) overrides the lettered block names and the stage lengths."""
	spec = [
		("conv", "conv1", 3, stem, 7, 2, 3, False),
		("bn", "bn_conv1", stem),
		("relu", "conv1_relu"),
		("maxpool", "pool1", 3, 2, 0),
	]

	inmaps = stem
	for level, (hmaps, nblocks) in enumerate(stages, start=2):
		names = blocknames[level] if blocknames is not None else ["%d%s" % (level, string.ascii_lowercase[b]) for b in range(nblocks)]
		for b, blockname in enumerate(names):
			spec += _resid(inmaps, hmaps, (1 if level == 2 else 2) if b == 0 else 1, blockname, b == 0)
			inmaps = 4 * hmaps

	spec += [("avgpool", "pool5", 7, 1, 0), ("flatten", "flatten"), ("linear", "fc%d" % classes, inmaps, classes)]
	if softmax:
		spec.append(("softmax", "prob"))
	return spec


def resnet50_spec(softmax=True):
	return resnet_spec(softmax=softmax)


def spec_param_shapes(spec, out=None):
	"""{"<name>.W": shape, ...} in spec order; BN running stats are listed under attrs."""
	params, attrs = out if out is not None else ({}, {})

	for layer in spec:
		kind = layer[0]

		if kind == "conv":
			_, name, cin, cout, size, _, _, bias = layer
			params[name + ".W"] = (cout, cin, size, size)
			if bias:
				params[name + ".b"] = (1, cout, 1, 1)

		elif kind == "bn":
			_, name, maps = layer
			params[name + ".scale"] = (1, maps, 1, 1)
			params[name + ".bias"] = (1, maps, 1, 1)
			attrs[name + ".mean"] = (1, maps, 1, 1)
			attrs[name + ".var"] = (1, maps, 1, 1)

		elif kind == "linear":
			_, name, nin, nout = layer
			params[name + ".W"] = (nin, nout)
			params[name + ".b"] = (nout, )

		elif kind == "resid":
			spec_param_shapes(layer[1], (params, attrs))
			spec_param_shapes(layer[2], (params, attrs))

	return params, attrs


def spec_out_shape(spec, shape):
	"""Walks dataShapeFrom (Modules/Conv2D.py:40-53, Modules/Pool2D.py:19-30) over a spec."""
	for layer in spec:
		kind = layer[0]

		if kind == "conv":
			_, _, _, cout, size, stride, pad, _ = layer
			n, _, h, w = shape
			shape = (n, cout, (h + 2 * pad - size) // stride + 1, (w + 2 * pad - size) // stride + 1)

		elif kind in ("maxpool", "avgpool"):
			_, _, size, stride, pad = layer
			n, c, h, w = shape
			shape = (n, c, (h + 2 * pad - size) // stride + 1, (w + 2 * pad - size) // stride + 1)

		elif kind == "flatten":
			n = shape[0]
			m = 1
			for d in shape[1:]:
				m *= d
			shape = (n, m)

		elif kind == "linear":
			shape = (shape[0], layer[3])

		elif kind == "resid":
			shape = spec_out_shape(layer[1], shape)

	return shape


# ------------------------------------------------------------------------------------------------
# builders (device networks on the MI355X backend)
# ------------------------------------------------------------------------------------------------

def build(spec, name=None, initscheme=None, wscale=1.0, actInplace=False, bnInplace=False):
	"""A spec as an executable network (puzzlelib_amd/engine.py). Parameters are drawn in spec order with the reference's
	RNG calls, so a numpy seed reproduces the reference builders' initial values."""
	from puzzlelib_amd.engine import Net
	return Net(spec, name=name, initscheme=initscheme, wscale=wscale, actInplace=actInplace, bnInplace=bnInplace)


def loadLeNet(modelpath=None, initscheme="none", name="lenet-5-like"):
	"""Models/Nets/LeNet.py:13-33; `modelpath`: a file the reference's Module.save wrote (puzzlelib_amd/checkpoint.py)."""
	net = build(lenet_spec(), name=name, initscheme=initscheme)
	if modelpath is not None:
		from puzzlelib_amd import checkpoint
		checkpoint.load(net, modelpath)
	return net


def buildNiN():
	"""TestLib/CnnCifar10NIN.py:13-49"""
	return build(nin_spec(), name="cifar", initscheme="gaussian", wscale=0.05)


def loadResNet(modelpath=None, layers="50", actInplace=False, bnInplace=False, initscheme="none", name=None):
	"""Models/Nets/ResNet.py:69-121: ResNet-50 / -101 / -152 (3-4-6-3, 3-4-23-3, 3-8-36-3 bottleneck blocks). `modelpath`: a file
	the reference's Module.save wrote, read the way ResNet.py:118-119 does (net.load(modelpath, assumeUniqueNames=True))."""
	names = resnet_blocknames(layers)
	stages = tuple((hmaps, len(names[level])) for level, hmaps in zip((2, 3, 4, 5), (64, 128, 256, 512)))

	net = build(
		resnet_spec(stages, blocknames=names), name="ResNet-%s" % layers if name is None else name, initscheme=initscheme,
		actInplace=actInplace, bnInplace=bnInplace
	)
	if modelpath is not None:
		from puzzlelib_amd import checkpoint
		checkpoint.load(net, modelpath)
	return net


def namedVariables(net):
	"""{"<layer name>.<param>": Param} — keys as used by the specs / the oracle."""
	return net.namedParams()


def namedAttrs(net):
	"""{"<layer name>.<attr>": GPUArray} (batch-norm running mean / var)."""
	return net.namedAttrs()

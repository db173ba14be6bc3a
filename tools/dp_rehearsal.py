"""Data-parallel rehearsal on ONE device: N ranks (python -m torch.distributed.run --nproc-per-node N) all bound to
device PUZZLE_MI355_DEVICE train the mini-ResNet on THE SAME batch through the full data-parallel path (parameter
broadcast, overlapped bucketed gradient exchange, 1/N scaling). RCCL refuses two ranks on one device, so the exchange
runs on the host-staged TCP fallback — everything around the transport (hooks, buckets, events, ordering) is the code
the multi-GPU run uses. With identical shards the mean gradient equals the single-process gradient bit for bit
((g + g) / 2 is exact), so rank 0's parameters must equal a single-process run's: tests/test_gpu_2_boundary.py checks that.

    python tools/dp_rehearsal.py OUT.npz            (single process)      or with RANK / WORLD_SIZE / MASTER_* set per rank"""
import os, sys
import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)

from puzzlelib_amd.settings import Config
from puzzlelib_amd import grid

out = sys.argv[1]
Config.deviceIdx = int(os.environ.get("PUZZLE_MI355_DEVICE", os.environ.get("LOCAL_RANK", "0")))
nodeinfo = grid.nodeFromEnv(bucketBytes=int(os.environ.get("PUZZLE_MI355_REHEARSE_BUCKET", 64 << 10)))          # small buckets: several exchanges overlap with backward

from puzzlelib_amd import nets, optim
from puzzlelib_amd.surface import bound

gpuarray = bound().gpuarray
golden = np.load(os.path.join(ROOT, "tests", "golden", "miniresnet.npz"))
spec = nets.resnet_spec(stages=((8, 1), (16, 2)), classes=10, stem=8, softmax=False)
spec = [l if l[0] != "avgpool" else ("avgpool", l[1], 8, 1, 0) for l in spec]

rank = int(os.environ.get("RANK", "0"))
np.random.seed(7 + 100 * rank)                              # different initial parameters per rank: the broadcast must fix that
net = nets.build(spec, name="mini", initscheme="he", actInplace=True)
if rank == 0:
	for name, var in net.namedParams().items():
		var.data.set(golden["init_" + name])

# PUZZLE_MI355_REHEARSE_AUTO=1: what an unpatched PuzzleLib does — the arena in sorted-name order (Optimizers/Optimizer.py:66-68),
# no hook into backward, only nodeinfo.sumTensor at update time: the overlap then comes from the arena's watcher (grid.ArenaWatcher)
auto = os.environ.get("PUZZLE_MI355_REHEARSE_AUTO", "0") == "1"
if auto:
	optim.Optimizer.arenaLayout = "sorted"
optimizer = optim.Adam(alpha=1e-3, nodeinfo=nodeinfo)
optimizer.setupOn(net, useGlobalState=True)
if nodeinfo is not None and not auto:
	grid.enableOverlap(optimizer, nodeinfo)

trainer = optim.Trainer(net, optim.CrossEntropy(), optimizer, batchsize=4)
data, labels = gpuarray.to_gpu(golden["data"]), gpuarray.to_gpu(golden["labels"])
for _ in range(int(os.environ.get("PUZZLE_MI355_REHEARSE_STEPS", "3"))):
	trainer.train(data, labels, random=False)

if rank == 0:
	watcher = nodeinfo.watcherOf("grad") if hasattr(nodeinfo, "watcherOf") else None
	np.savez(out, transport=np.array(getattr(nodeinfo, "transport", "single")),
			 auto_buckets=np.array(len(watcher.reducer.buckets) if watcher is not None and watcher.reducer is not None else 0),
			 auto_ranges=np.array(sum(len(b.ranges) for b in watcher.reducer.buckets) if watcher is not None and watcher.reducer is not None else 0),
			 **{name: var.data.get() for name, var in net.namedParams().items()})
grid.barrier()

#!/usr/bin/env python3
"""
Removes the measurement rig (timing-only ablation branches, experiment switches) from the shipped kernel sources.

The kernels grew compile-time switches while they were tuned: PZ_ABL / WN_ABL / W4_ABL bits ("wrong results, timing only"),
PZ_IG_PRIO, PZ_EPI_AUX, W4_DUMMY_VALU ... The shipped library must not be able to contain them. This script evaluates every
preprocessor conditional that tests ONLY rig macros at the shipped values (below) and keeps the live branch; conditionals
on anything else are left alone. The reverse diff is written to tools/dev/measurement_rig.patch: `git apply` it onto a
scratch copy of csrc/ to get the instrumented sources back (tools/ablate.sh does that).

usage: python tools/dev/strip_rig.py            (rewrites puzzlelib_amd/csrc/{conv,wino,wino4,thin}.hip in place)
"""
import os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "puzzlelib_amd", "csrc")

# shipped values; None = not defined
RIG = {
	"PZ_ABL": 0, "WN_ABL": 0, "W4_ABL": 0, "PZ_IG_PRIO": 0, "PZ_EPI_AUX": 0,
	"PZ_EPI_FORCE_SCALAR": None, "W4_DUMMY_VALU": None, "W4_PRIO": None, "W4_NOSB": None, "PZ_THIN_ABL": None,
}
# object-like / function-like rig macros that are substituted in the code itself
SUBST = [
	(re.compile(r"PZ_ABL_NEAR\(((?:[^()]|\([^()]*\))*)\)"), r"\1"),
	(re.compile(r",\s*PZ_EPI_AUX\)"), ", 0)"),
]


def evaluate(kind, expr):
	"""value of a conditional that mentions rig macros only, else None"""
	expr = expr.split("//")[0].strip()
	if kind in ("ifdef", "ifndef"):
		if expr not in RIG:
			return None
		defined = RIG[expr] is not None
		return defined if kind == "ifdef" else not defined
	names = set(re.findall(r"[A-Za-z_]\w*", expr)) - {"defined"}
	if not names or not names <= set(RIG):
		return None
	py = re.sub(r"defined\s*\(?\s*(\w+)\s*\)?", lambda m: "1" if RIG[m.group(1)] is not None else "0", expr)
	py = re.sub(r"[A-Za-z_]\w*", lambda m: str(RIG[m.group(0)] or 0), py)
	py = py.replace("&&", " and ").replace("||", " or ").replace("!", " not ")
	return bool(eval(py))


def strip(text):
	out = []
	# stack entries: [mode, taken, emitting]   mode: "rig" (resolved here) or "keep" (left to the compiler)
	stack = []

	def live():
		return all(e[2] for e in stack)

	lines = text.split("\n")
	i = 0
	while i < len(lines):
		line = lines[i]
		m = re.match(r"\s*#\s*(ifdef|ifndef|if|elif|else|endif)\b(.*)", line)
		if not m:
			# a rig macro's own default definition:  #ifndef X / #define X v / (comment lines) / #endif  is handled as a conditional
			if live():
				for rx, rep in SUBST:
					line = rx.sub(rep, line)
				out.append(line)
			i += 1
			continue

		kind, expr = m.group(1), m.group(2)
		if kind in ("if", "ifdef", "ifndef"):
			val = evaluate(kind, expr)
			if val is None:
				stack.append(["keep", False, True])
				if live():
					out.append(line)
			else:
				stack.append(["rig", val, val])
		elif kind == "elif":
			top = stack[-1]
			if top[0] == "keep":
				if live():
					out.append(line)
			else:
				val = evaluate("if", expr)
				assert val is not None, "rig #if with a non-rig #elif: " + line
				top[2] = (not top[1]) and val
				top[1] = top[1] or val
		elif kind == "else":
			top = stack[-1]
			if top[0] == "keep":
				if live():
					out.append(line)
			else:
				top[2] = not top[1]
				top[1] = True
		else:
			top = stack.pop()
			if top[0] == "keep" and live():
				out.append(line)
		i += 1

	assert not stack
	return "\n".join(out)


def drop_default_blocks(text):
	"""`#ifndef X \n #define X 0 ... \n #endif` of a rig macro evaluates to its #define staying: remove the define too"""
	for name in RIG:
		text = re.sub(r"^#define %s\b.*\n(?:[ \t]+//.*\n)*" % name, "", text, flags=re.M)
	return text


def main():
	changed = []
	for fn in ("conv.hip", "wino.hip", "wino4.hip", "thin.hip"):
		path = os.path.join(CSRC, fn)
		src = open(path).read()
		new = drop_default_blocks(strip(src))
		if new != src:
			open(path, "w").write(new)
			changed.append(fn)
	print("stripped:", changed)
	diff = subprocess.run(["git", "diff", "-R", "--", "puzzlelib_amd/csrc"], cwd=ROOT, capture_output=True, text=True).stdout
	if diff.strip():
		open(os.path.join(ROOT, "tools", "dev", "measurement_rig.patch"), "w").write(diff)
		print("wrote tools/dev/measurement_rig.patch (%d lines)" % diff.count("\n"))


if __name__ == "__main__":
	main()

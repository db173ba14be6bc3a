#!/usr/bin/env python3
"""Registers / scratch / LDS / occupancy of the kernels in a gfx950 assembly listing (hipcc -save-temps=obj ... or
--cuda-device-only -S), optionally filtered by a substring of the demangled-ish name.
    python tools/kernel_regs.py /tmp/conv-hip-amdgcn-amd-amdhsa-gfx950.s igemm_persist"""
import re, subprocess, sys

text = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"^(_Z\S+):\s.*?\n(.*?)^\s*\.end_amdhsa_kernel", text, re.M | re.S):
	name, body = m.group(1), m.group(2)
	if pat not in name:
		continue
	tail = text[m.end():m.end() + 4000]
	get = lambda key: (re.search(r"; %s: (\d+)" % key, tail) or [None, "?"])[1]
	try:
		dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
	except Exception:
		dem = name
	dem = re.sub(r"\(anonymous namespace\)::", "", dem)
	print("%-90s vgpr %s agpr %s scratch %s lds %s occupancy %s" % (dem[:90], get("NumVgprs"), get("NumAgprs"), get("ScratchSize"), get("LDSByteSize"), get("Occupancy")))

#!/usr/bin/env python3
"""Writes a copy of a cascade XML keeping only the first N stages (kernel-timing experiments)."""
import sys
import xml.etree.ElementTree as ET

src, n, dst = sys.argv[1], int(sys.argv[2]), sys.argv[3]
tree = ET.parse(src)
casc = list(tree.getroot())[0]
stages = casc.find("stages")
for s in [s for s in stages if s.tag == "_"][n:]:
    stages.remove(s)
casc.find("stageNum").text = str(n)
tree.write(dst)

#!/bin/bash
cd /root/repo
python profiles/pcie_probe.py 2>/dev/null | tail -1
python profiles/pcie_probe.py --torch-first 2>/dev/null | tail -1

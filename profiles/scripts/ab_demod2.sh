# ab_demod.sh + the parity tests most sensitive to the demodulator, for the product build only
cd /root/repo
bash profiles/scripts/ab_demod.sh "$@"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3

O=gpurun_out/r02_x; mkdir -p $O
python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log | cut -c1-200

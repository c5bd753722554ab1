O=gpurun_out/r02_g; mkdir -p $O
python3 -m pytest tests -m gpu -x -q -k "spmv or lsmr or iteration or errors or sharded" > $O/pytest_a.log 2>&1; tail -15 $O/pytest_a.log
python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -60 $O/pytest.log

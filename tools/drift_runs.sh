for i in 1 2 3; do
python tools/step_drift.py 300 1 0
python tools/step_drift.py 300 6 0
python tools/step_drift.py 300 1 500
done

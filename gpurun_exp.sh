export SA_GUARD=0
(time python -m pytest tests -m gpu -q -x --timeout 1500 -k "mem or rn129 or rn22_1 or every_mapping or mapping" 2>&1 | tail -8) 2>&1 | tail -14

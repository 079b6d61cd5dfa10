for r in 4 8 12 16 24; do echo "== KC_POLY_ROWS=$r"; KC_POLY_ROWS=$r MODES=1 bash profiles/down_ab.sh 2>/dev/null | grep poly; done

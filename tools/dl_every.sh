for n in 1 2 3 100; do echo "DL_EVERY=$n"; SPARTACUS_HOST_DL_EVERY=$n timeout -k 10 200 python tools/host_entry_modes.py 2>&1 | tail -n 3; done

for nb in 0 1; do for k in 16384 65536 262144; do
echo "NOBOUNDS $nb K $k: $(SLAMHIP_K1_NOBOUNDS=$nb timeout 120 python tools/k1exp.py $k 2>&1 | grep -o 'dist [0-9.]* us\|selfcheck [0-9]*' | tr '\n' ' ')"
done; done

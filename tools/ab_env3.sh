# same-box A/B of one environment variable, three interleaved pairs: bash tools/ab_env3.sh VAR A B '<command>'   (A B A B A B)
for v in "$2" "$3" "$2" "$3" "$2" "$3"; do echo "== $1=$v"; env "$1=$v" bash -c "$4"; done

bash tools/profile_round4.sh r4_t

"""pilco_amd: MI355X-native PILCO moment-matching path."""

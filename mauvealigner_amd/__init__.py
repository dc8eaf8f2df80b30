"""mauvealigner_amd -- MI355X-native hot path of mauveAligner/progressiveMauve."""

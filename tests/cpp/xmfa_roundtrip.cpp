// Reads an XMFA file with the mirror, loads the sequences its header names, writes it back: exit code 0 when the text is
// byte-identical.  tests/test_compat_headers.py runs it on XMFA written by the reference's own src/mfa2xmfa.cpp.
#include <cassert>
#include <fstream>
#include <iostream>
#include <sstream>
#include "libMems/mems_hip.h"
using namespace mems;
int main(int argc, char **argv)
{
    std::ifstream in(argv[1]); std::stringstream want; want << in.rdbuf();
    std::istringstream is(want.str());
    IntervalList il; il.ReadStandardAlignment(is);
    MatchList ml; ml.seq_filename = il.seq_filename;
    if (ml.seq_filename.size() > 1 && ml.seq_filename[0] == ml.seq_filename[1]) LoadMFASequences(ml, ml.seq_filename[0], nullptr); else LoadSequences(ml, nullptr);
    il.seq_table = ml.seq_table;
    std::ostringstream os; il.WriteStandardAlignment(os);
    std::cout << os.str();
    return os.str() == want.str() ? 0 : 1;
}

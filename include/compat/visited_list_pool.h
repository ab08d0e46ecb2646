// Forwarding header: a harness written against the reference's include/visited_list_pool.h (tests/test_search_roargraph.cpp:14-18)
// compiles unchanged with -I<this repo>/include/compat; everything it used from there is in the facade.
#pragma once
#include "../roargraph_amd/efanna2e.h"

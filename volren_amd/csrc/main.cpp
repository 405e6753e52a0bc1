// main.cpp -- `volren` offline CLI (placeholder until the parse_cmd row of SURVEY 8f-1 is built out)
#include <cstdio>
int main() { std::puts("volren (HIP): use the C ABI / python binding; CLI offline driver lands with SURVEY 8f-1"); return 0; }

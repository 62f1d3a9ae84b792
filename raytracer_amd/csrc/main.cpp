// Command line of the drop-in: `RayTracer <config.txt>`, the contract of /root/reference/src/main.cpp:4-16.
// Exactly one argument, else "Invalid arguments" and -1; a failure while parsing or rendering is
// printed and the process still exits with 0, as the reference does.  The environment variable
// RTX_DEVICES selects how many GPUs share the image (default 1).
#include <cstdlib>
#include <iostream>

#include "../../include/rtx.h"

int main(int args, char** argv) {
	if (args != 2) {
		std::cout << "Invalid arguments\n";
		return -1;
	}
	int devices = 1;
	if (const char* e = std::getenv("RTX_DEVICES")) devices = std::atoi(e);
	if (rtxh_run_config(argv[1], "", devices) < 0) std::cout << rtxh_last_error() << std::endl;
	return 0;
}

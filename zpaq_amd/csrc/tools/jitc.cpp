// zpq_jitc -- compiles one generated kernel source with hipRTC in a process of its own.
//
//   zpq_jitc <source file> <code object file> <hipRTC option>...
//
// hipRTC serialises compilations inside one process (four threads compile four kernels in the time of four), so the
// library's spec_precompile() runs several of these side by side when a batch brings several headers nobody has
// compiled yet (device/spec_loader.cpp).  Needs no GPU.  Exit status 0 = the code object was written; otherwise the
// compiler's log is on stderr.
#include <hip/hiprtc.h>

#include <cstdio>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

int main(int argc, char** argv) {
  if (argc < 3) {
    fprintf(stderr, "usage: zpq_jitc <source file> <code object file> <hipRTC option>...\n");
    return 2;
  }
  std::ifstream f(argv[1], std::ios::binary);
  if (!f) { fprintf(stderr, "zpq_jitc: cannot read %s\n", argv[1]); return 2; }
  std::ostringstream ss;
  ss << f.rdbuf();
  const std::string source = ss.str();
  hiprtcProgram prog;
  if (hiprtcCreateProgram(&prog, source.c_str(), "zpq_spec.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
    fprintf(stderr, "zpq_jitc: hiprtcCreateProgram failed\n");
    return 1;
  }
  const hiprtcResult r = hiprtcCompileProgram(prog, argc - 3, (const char**)(argv + 3));
  size_t ls = 0;
  if (hiprtcGetProgramLogSize(prog, &ls) == HIPRTC_SUCCESS && ls > 1) {
    std::string log(ls, '\0');
    hiprtcGetProgramLog(prog, &log[0]);
    fprintf(stderr, "%s\n", log.c_str());
  }
  if (r != HIPRTC_SUCCESS) return 1;
  size_t cs = 0;
  hiprtcGetCodeSize(prog, &cs);
  std::vector<char> code(cs);
  hiprtcGetCode(prog, code.data());
  hiprtcDestroyProgram(&prog);
  const std::string part = std::string(argv[2]) + ".part";      // renamed into place only when complete
  std::ofstream o(part, std::ios::binary);
  o.write(code.data(), (std::streamsize)code.size());
  o.close();
  if (!o || std::rename(part.c_str(), argv[2]) != 0) { std::remove(part.c_str()); return 2; }
  return 0;
}

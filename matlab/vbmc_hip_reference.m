function f = vbmc_hip_reference(name)
%VBMC_HIP_REFERENCE Handle to the reference implementation shadowed by a shim of the same name.
p = which(name,'-all');
here = fileparts(mfilename('fullpath'));
for i = 1:numel(p)
    if ~strncmp(p{i},here,numel(here))
        old = cd(fileparts(p{i})); c = onCleanup(@() cd(old));
        f = str2func(name);  % resolved while the reference folder is the current folder
        return;
    end
end
error('vbmc_hip:noreference','No reference implementation of %s found on the path.',name);
end
